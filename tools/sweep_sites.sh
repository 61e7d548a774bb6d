#!/bin/bash
# Window-length sweep of the K = 69 kernel on full waves (1000 haplotypes, all pairs): per-site cost against the
# window length, single-chunk and chunked.  Output: gpurun_out/sweep_sites.txt
set -e
out=gpurun_out/sweep_sites.txt
: > $out
for s in 256 384 512 768 1024 1536 2048 3000 4096 8192; do
  python bench.py --haps 1000 --sites $s --steps 5 --warmup 2 --cpu-pairs 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d['config']
print($s, d['ms_per_step'], d['roofline']['frac'], c.get('kernel'), c.get('chunk_sites'), c.get('beta_stride'), c.get('resident_chunks'))
" >> $out
done
cat $out
