#!/bin/bash
# Runs on the GPU box: the measurements of round 4 that are kept under profiles/ (stamped by tools/stamp_profiles.py).
# Usage: tools/r04_measure.sh <part>   part = headline | pmc | round | all
set -u
cd $GRAFT_REPO_ROOT
PART=${1:-all}
if [ $PART = headline ] || [ $PART = all ]; then
  # 1. the headline command under rocprofv3 --kernel-trace --stats and the two HBM-traffic counters (separate passes)
  bash tools/profile_bench.sh r04 --steps 3 --warmup 1 > gpurun_out/r04_headline.log 2>&1
  # ... and the default line itself (other workloads + the library-default workspace figure), untraced
  python3 bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
  echo "headline done"
fi
if [ $PART = pmc ] || [ $PART = all ]; then
  # 2. counter sets AT SIZE: the C2 workload itself (K = 69 kernel) and C4 (K = 256 wave-group kernel)
  bash tools/prof_counters.sh r04_c2_at_size > gpurun_out/r04_pmc_c2.log 2>&1
  bash tools/prof_counters.sh r04_c4_at_size --workload c4 > gpurun_out/r04_pmc_c4.log 2>&1
  bash tools/prof_counters_cmd.sh r04_identify id_ tools/measure_configs.py identify > gpurun_out/r04_pmc_identify.log 2>&1
  echo "pmc done"
fi
if [ $PART = round ] || [ $PART = all ]; then
  bash tools/measure_round.sh r04 > gpurun_out/r04_round.log 2>&1
  python3 tools/measure_configs.py k320 k350 k402 k448 k500 k600 > gpurun_out/round_r04/wide_beyond_256.jsonl 2> gpurun_out/round_r04/wide_beyond_256.err
  echo "round done"
fi
