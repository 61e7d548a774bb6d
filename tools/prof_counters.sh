#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of a reduced bench workload.
# Usage: tools/prof_counters.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --no-other-workloads --steps 1 --warmup 0 --cpu-pairs 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  N=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$N -- python3 $ARGS > $OUT/pmc_$N.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
out="$OUT"
tot=collections.defaultdict(float)
for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "decode_kernel" in r.get("Kernel_Name",""):
            tot[r["Counter_Name"]]+=float(r["Counter_Value"])
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from fastsmc_amd.build import hip_source_hash
with open(out+"/summary.txt","w") as w:
    w.write(f"lib_hash\t{hip_source_hash()}\n")
    w.write("bench_args\t$*\n")
    names=set()
    for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "decode_kernel" in r.get("Kernel_Name",""): names.add(r["Kernel_Name"])
    for n in sorted(names): w.write(f"kernel\t{n}\n")
    for k in sorted(tot): w.write(f"{k}\t{tot[k]:.6g}\n")
    for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
        w.write(open(f).read())
print(open(out+"/summary.txt").read())
PY
