#!/bin/bash
# Here (no GPU): everything tools/r05_measure.sh left under gpurun_out/ into profiles/, stamped with the library's hash.
# Usage: tools/stamp_round.sh [tag]      (default r05)
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r05}
python3 tools/stamp_profiles.py traffic gpurun_out/profile_$TAG $TAG | grep -E "kernel_avg|events|hbm_bytes|lib_hash"
for n in c2_at_size c4_at_size identify c1_lone_wave; do python3 tools/stamp_profiles.py pmc gpurun_out/prof_${TAG}_$n ${TAG}_pmc_$n; done
python3 tools/stamp_profiles.py round gpurun_out/round_$TAG $TAG | tail -1
cp gpurun_out/round_$TAG/wide_beyond_256.jsonl profiles/${TAG}_wide_beyond_256_states.jsonl
cp gpurun_out/${TAG}_bench_default.json profiles/${TAG}_bench_default.json
cp gpurun_out/round_$TAG/c1_consumers.jsonl profiles/${TAG}_c1_consumers.jsonl
cp gpurun_out/round_$TAG/malloc_cost.txt profiles/${TAG}_malloc_cost.txt
if [ -f gpurun_out/round_$TAG/run_c1_asmc.jsonl ]; then cp gpurun_out/round_$TAG/run_c1_asmc.jsonl profiles/${TAG}_run_c1_asmc_published_job.jsonl; fi
cp gpurun_out/round_$TAG/c1_sums_bench.json profiles/${TAG}_c1_sums_bench_line.json
(cat gpurun_out/round_$TAG/run_c2_timeline.json; grep fsmc gpurun_out/round_$TAG/run_c2_timeline.err) > profiles/${TAG}_run_c2_timeline.txt
python3 tools/resource_table.py > /tmp/resource_table.log 2>&1; tail -1 /tmp/resource_table.log | cut -c1-160
