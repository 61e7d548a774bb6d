#!/bin/bash
# usage: prof_var.sh <variant> ; runs on the GPU box
V=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pv_$V
mkdir -p $OUT
export FSMC_HIP_LIB=$GRAFT_REPO_ROOT/fastsmc_amd/variants/lib$V.so
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --workload c4 --sites 40000 --steps 1 --warmup 0 --cpu-pairs 0 --no-other-workloads"
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 120 rocprofv3 --pmc $SET --output-format csv -d $OUT/pmc_$N -- python3 $ARGS > $OUT/pmc_$N.log 2>&1
done
python3 - <<PY
import csv, glob, collections
out="$OUT"
tot=collections.defaultdict(float)
for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "decode_kernel" in r.get("Kernel_Name",""):
            tot[r["Counter_Name"]]+=float(r["Counter_Value"])
print("== $V")
for k in sorted(tot): print(f"{k}\t{tot[k]:.6g}")
PY
