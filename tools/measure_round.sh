#!/bin/bash
# Runs on the GPU box: the side measurements of a round that are kept under profiles/ next to the headline profile
# (tools/profile_bench.sh): C3 single-GPU point, C4 at size with rocprofv3 kernel stats + traffic, the short-window
# (hashing) regime with kernel stats, the padded family members, the identification step with kernel stats, the product
# path end to end (with and without the hashing pre-filter).
# Usage: tools/measure_round.sh <tag>
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/round_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py --workload c3 --steps 2 --warmup 1 --cpu-pairs 0 > $OUT/c3_n1.json 2> $OUT/c3_n1.err
echo "c3 done"
bash tools/profile_bench.sh ${TAG}_c4 --states 256 --haps 256 --sites 200000 --steps 1 --warmup 1 > $OUT/c4_profile.log 2>&1
echo "c4 done"
python3 bench.py --states 256 --haps 600 --sites 3000 --steps 2 --warmup 1 --cpu-pairs 0 > $OUT/c4_reduced.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/short_trace -- python3 $GRAFT_REPO_ROOT/tools/measure_configs.py short > $OUT/short.json 2> $OUT/short.err )
cp $(find $OUT/short_trace -name "*kernel_stats.csv" | head -1) $OUT/short_kernel_stats.csv 2>/dev/null
echo "short done"
: > $OUT/family.jsonl
for K in 16 32 48 50 64 80 96 100 112 128 192; do
  python3 bench.py --steps 2 --warmup 1 --cpu-pairs 0 --states $K > $OUT/fam.json 2>/dev/null && cat $OUT/fam.json >> $OUT/family.jsonl
done
echo "family done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/identify_trace -- python3 $GRAFT_REPO_ROOT/tools/measure_configs.py identify > $OUT/identify.jsonl 2> $OUT/identify.err )
cp $(find $OUT/identify_trace -name "*kernel_stats.csv" | head -1) $OUT/identify_kernel_stats.csv 2>/dev/null
python3 tools/measure_configs.py hashing > $OUT/hashing.json 2> $OUT/hashing.err
python3 tools/measure_configs.py seq seq100 > $OUT/sequence.jsonl 2> $OUT/sequence.err
python3 tools/measure_configs.py k256 k192 k300 > $OUT/wide_w2.jsonl 2> $OUT/wide_w2.err
echo "identify done"
python3 tools/measure_configs.py run_c2 > $OUT/run_c2.json 2> $OUT/run_c2.err
python3 tools/measure_configs.py c1 > $OUT/c1.json 2> $OUT/c1.err
python3 bench.py --workload c1 --steps 10 --warmup 2 --cpu-pairs 0 > $OUT/c1_bench.json 2> $OUT/c1_bench.err
python3 tools/measure_configs.py c5_job > $OUT/c5_job.json 2> $OUT/c5_job.err
python3 tools/measure_configs.py ingest_c3 > $OUT/ingest_c3.json 2> $OUT/ingest_c3.err
echo "host done"
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "all done"
