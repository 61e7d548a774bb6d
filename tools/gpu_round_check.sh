#!/bin/bash
# Runs on the GPU box: the whole GPU test suite, then the default bench line (what the driver runs at round end).
# Usage: tools/gpu_round_check.sh <tag>
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/check_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
free -g > $OUT/host_memory.txt 2>&1; nproc >> $OUT/host_memory.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
RC=$?
echo "pytest rc=$RC"; tail -5 $OUT/pytest_gpu.log
if [ $RC -ne 0 ]; then exit $RC; fi
timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; tail -c 6000 $OUT/bench_default.json
