#!/bin/bash
# Runs on the GPU box: the wave-group kernel's members against each other on one box -- parity tests of the wide models
# under a forced member (FSMC_DIAG_W2_MEMBER), then the 600 x 3000 list and C4 at size with each member.
# Usage: tools/ab_w2_member.sh <tag> <member> [<member> ...]     e.g. tools/ab_w2_member.sh r05a 2x128 4x64
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for M in "$@"; do
  FSMC_DIAG_W2_MEMBER=$M timeout -k 10 600 python3 -m pytest tests/test_gpu_wide_model.py tests/test_gpu_generic_k.py tests/test_gpu_sequence.py -x -q -m gpu -k "wide or 130 or 192 or 200 or 256 or seq" > $OUT/tests_$M.log 2>&1
  echo "tests $M rc=$?"; tail -3 $OUT/tests_$M.log
done
for M in "$@"; do
  for rep in 1 2; do
    FSMC_DIAG_W2_MEMBER=$M timeout -k 10 300 python3 bench.py --states 256 --haps 600 --sites 3000 --steps 3 --warmup 1 --cpu-pairs 0 >> $OUT/reduced_$M.jsonl 2>> $OUT/reduced_$M.err || echo "reduced $M failed"
  done
done
for M in "$@"; do
  FSMC_DIAG_W2_MEMBER=$M timeout -k 10 300 python3 bench.py --workload c4 --steps 1 --warmup 1 --cpu-pairs 0 >> $OUT/c4_$M.jsonl 2>> $OUT/c4_$M.err || echo "c4 $M failed"
done
python3 - $OUT "$@" <<'PY'
import json, sys, glob
out = sys.argv[1]
for m in sys.argv[2:]:
    for kind in ("reduced", "c4"):
        try:
            for ln in open(f"{out}/{kind}_{m}.jsonl"):
                d = json.loads(ln)
                print(m, kind, "member", d["config"]["kernel_member"], "kernel_ms %.1f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"], "records", d["config"]["ibd_records_per_step"])
        except Exception as e:
            print(m, kind, "no result", e)
PY
