#!/bin/bash
# Runs on the GPU box: instruction-cache / scalar-cache PMC passes of a reduced bench workload.
# Usage: tools/prof_icache.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/icache_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-pairs 0 $*"
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
  N=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$N -- python3 $ARGS > $OUT/pmc_$N.log 2>&1
done
python3 - <<PY
import csv, glob, collections
out="$OUT"
tot=collections.defaultdict(float)
for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "decode_kernel" in r.get("Kernel_Name",""):
            tot[r["Counter_Name"]]+=float(r["Counter_Value"])
with open(out+"/summary.txt","w") as w:
    for k in sorted(tot): w.write(f"{k}\t{tot[k]:.6g}\n")
print(open(out+"/summary.txt").read())
PY
