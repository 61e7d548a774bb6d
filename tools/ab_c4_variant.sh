#!/bin/bash
# Runs on the GPU box: config 4 (and the 600 x 3000 list) with and without an environment switch, interleaved on one box.
# Usage: tools/ab_c4_variant.sh <ENV_NAME>     (base: unset; variant: ENV_NAME=1)
set -u
cd $GRAFT_REPO_ROOT
V=${1:-FSMC_DIAG_W2_GHOSTS}
for rep in 1 2; do
  for L in base variant; do
    if [ $L = variant ]; then export $V=1; else unset $V; fi
    timeout -k 10 300 python3 bench.py --workload c4 --steps 1 --warmup 1 --cpu-pairs 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L c4 kernel_ms %.1f frac %.4f records %d' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['ibd_records_per_step']))"
    timeout -k 10 300 python3 bench.py --states 256 --haps 600 --sites 3000 --steps 3 --warmup 1 --cpu-pairs 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L reduced kernel_ms %.1f frac %.4f records %d' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['ibd_records_per_step']))"
  done
done
