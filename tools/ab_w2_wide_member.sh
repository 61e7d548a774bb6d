#!/bin/bash
# Runs on the GPU box: members of the wave-group kernel for models of 257 ... 512 states against each other on one box
# (a variant library built by tools/build_variant.py --extra-w2 holds the candidates; FSMC_DIAG_W2_MEMBER picks one where
# it holds the model, otherwise the shipped choice runs) -- parity tests of the wide models, then the 600 x 3000 list.
# Usage: tools/ab_w2_wide_member.sh <tag> <variant .so> <member> [<member> ...]   e.g. ... r05w fastsmc_amd/variants/libw128.so 3x128 4x128 shipped
set -u
TAG=$1; LIB=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/abw_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export FSMC_HIP_LIB=$GRAFT_REPO_ROOT/$LIB
for M in "$@"; do
  FSMC_DIAG_W2_MEMBER=$M timeout -k 10 600 python3 -m pytest tests/test_gpu_wide_model.py tests/test_gpu_generic_k.py -q -m gpu > $OUT/tests_$M.log 2>&1
  echo "tests $M rc=$?"; tail -12 $OUT/tests_$M.log | cut -c1-300
done
for K in 350 384 402 448 500 512; do
  for M in "$@"; do
    FSMC_DIAG_W2_MEMBER=$M timeout -k 10 300 python3 bench.py --states $K --haps 600 --sites 3000 --steps 2 --warmup 1 --cpu-pairs 0 > $OUT/k${K}_$M.json 2> $OUT/k${K}_$M.err || echo "K=$K $M failed"
    python3 -c "
import json,sys
d=json.load(open('$OUT/k${K}_$M.json'))
print('K=$K', '$M', 'member', d['config']['kernel_member'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'], 'frac %.3f' % d['roofline']['frac'], 'records', d['config']['ibd_records_per_step'])" || true
  done
done
