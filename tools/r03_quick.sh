#!/bin/bash
# Runs on the GPU box: quick A/B of a build -- parity tests of the kernels that changed, then the short benches.
# Usage: tools/r03_quick.sh <tag> [tests...]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/q_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest -x -q -m gpu "$@" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
python3 bench.py --states 256 --haps 600 --sites 3000 --steps 2 --warmup 1 --cpu-pairs 0 > $OUT/c4_reduced.json 2> $OUT/c4_reduced.err
python3 bench.py --states 256 --haps 256 --sites 200000 --steps 1 --warmup 1 --cpu-pairs 0 > $OUT/c4_at_size.json 2> $OUT/c4_at_size.err
python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 > $OUT/c2.json 2> $OUT/c2.err
python3 - <<PY
import json
for n in ("c4_reduced","c4_at_size","c2"):
    try:
        d=json.load(open("$OUT/%s.json"%n))
        print(n, "kernel_ms", round(d["roofline"]["kernel_ms"],1), "frac", round(d["roofline"]["frac"],3), "pairs/s", round(d["value"]), d["config"].get("ibd_records_per_step"))
    except Exception as e:
        print(n, "FAILED", e)
PY
