#!/bin/bash
# Runs on the GPU box: config 4 at size and the 600 x 3000 list with each of the given variant libraries
# (fastsmc_amd/variants/lib<name>.so; "base" = the shipped library), interleaved, twice.
# Usage: tools/ab_libs.sh base <name> [<name> ...]
set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for L in "$@"; do
    if [ $L = base ]; then unset FSMC_HIP_LIB; else export FSMC_HIP_LIB=fastsmc_amd/variants/lib$L.so; fi
    timeout -k 10 300 python3 bench.py --workload c4 --steps 1 --warmup 1 --cpu-pairs 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L c4 kernel_ms %.1f frac %.4f records %d' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['ibd_records_per_step']))"
    timeout -k 10 300 python3 bench.py --states 256 --haps 600 --sites 3000 --steps 3 --warmup 1 --cpu-pairs 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L reduced kernel_ms %.1f frac %.4f records %d' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['ibd_records_per_step']))"
  done
done
