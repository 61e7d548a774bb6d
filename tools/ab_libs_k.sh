#!/bin/bash
# Runs on the GPU box: the 600 x 3000 list at the given numbers of states with each of the given variant libraries
# (fastsmc_amd/variants/lib<name>.so; "base" = the shipped library), interleaved, twice; then the wide models' parity
# tests with every variant.
# Usage: tools/ab_libs_k.sh "<K> [<K> ...]" base <name> [<name> ...]
set -u
cd $GRAFT_REPO_ROOT
KS=$1; shift
for rep in 1 2; do
  for K in $KS; do
    for L in "$@"; do
      if [ $L = base ]; then unset FSMC_HIP_LIB; else export FSMC_HIP_LIB=$GRAFT_REPO_ROOT/fastsmc_amd/variants/lib$L.so; fi
      timeout -k 10 300 python3 bench.py --states $K --haps 600 --sites 3000 --steps 2 --warmup 1 --cpu-pairs 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L K=$K member %d kernel_ms %.1f frac %.4f records %d' % (d['config']['kernel_member'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['ibd_records_per_step']))"
    done
  done
done
for L in "$@"; do
  if [ $L = base ]; then continue; fi
  export FSMC_HIP_LIB=$GRAFT_REPO_ROOT/fastsmc_amd/variants/lib$L.so
  timeout -k 10 600 python3 -m pytest tests/test_gpu_wide_model.py tests/test_gpu_generic_k.py tests/test_gpu_sequence.py -q -m gpu -k "wide or 2[0-9][0-9] or 3[0-9][0-9] or seq" 2>&1 | tail -3
done
