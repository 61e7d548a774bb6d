#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes (separate passes, no tracing with counters) of
# ANY command, summed over the kernels whose name contains <kernel-substring>.
# Usage: tools/prof_counters_cmd.sh <tag> <kernel-substring> <python script + args relative to the repo root>
set -u
TAG=$1; SUB=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/$*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$N -- python3 $ARGS > $OUT/pmc_$N.log 2>&1
done
python3 - <<PY
import csv, glob, collections, sys
out="$OUT"; sub="$SUB"
tot=collections.defaultdict(float); names=set()
for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r.get("Kernel_Name",""):
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); names.add(r["Kernel_Name"])
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from fastsmc_amd.build import hip_source_hash
with open(out+"/summary.txt","w") as w:
    w.write(f"lib_hash\t{hip_source_hash()}\n")
    w.write("command\t$*\n")
    for n in sorted(names): w.write(f"kernel\t{n}\n")
    for k in sorted(tot): w.write(f"{k}\t{tot[k]:.6g}\n")
    for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
        w.write(open(f).read())
print(open(out+"/summary.txt").read())
PY
