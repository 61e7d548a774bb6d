#!/bin/bash
# Runs on the GPU box: config 4 at size with explicit chunk lengths (resident chunks as the workspace holds), interleaved.
# Usage: tools/ab_c4_chunk.sh <chunk> [<chunk> ...]      (0 = the library's default)
set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for C in "$@"; do
    timeout -k 10 300 python3 bench.py --workload c4 --steps 1 --warmup 1 --cpu-pairs 0 --chunk-sites $C 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $C ->', d['config']['chunk_sites'], 'chunks', d['config']['chunks_per_window'], 'resident', d['config']['resident_chunks'], 'kernel_ms %.1f frac %.4f records %d' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['ibd_records_per_step']))"
  done
done
