#!/bin/bash
# Runs on the GPU box: the models beyond 256 states -- parity tests, then the 600 x 3000 list at a few K.
# Usage: tools/gpu_wide_check.sh <tag> [K ...]      (FSMC_HIP_LIB is honoured: a variant library)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/wide_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_generic_k.py tests/test_gpu_wide_model.py tests/test_gpu_sequence.py -q -m gpu > $OUT/pytest.log 2>&1
RC=$?
echo "pytest rc=$RC"; tail -15 $OUT/pytest.log | cut -c1-300
for K in "$@"; do
  timeout -k 10 300 python3 bench.py --states $K --haps 600 --sites 3000 --steps 2 --warmup 1 --cpu-pairs 0 > $OUT/k$K.json 2> $OUT/k$K.err || echo "K=$K failed"
  python3 -c "
import json
d=json.load(open('$OUT/k$K.json'))
print('K=$K member', d['config']['kernel_member'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], 'frac %.3f' % d['roofline']['frac'], 'records', d['config']['ibd_records_per_step'])" || true
done
