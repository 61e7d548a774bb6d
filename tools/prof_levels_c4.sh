#!/bin/bash
# Runs on the GPU box: in-flight "level" counters of the wave-group kernel (average latency per instruction class =
# LEVEL / INSTS).  Usage: tools/prof_levels_c4.sh [variant]
V=${1:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/levels_${V:-shipped}
mkdir -p $OUT
if [ -n "$V" ]; then export FSMC_HIP_LIB=$GRAFT_REPO_ROOT/fastsmc_amd/variants/lib$V.so; fi
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --workload c4 --sites 40000 --steps 1 --warmup 0 --cpu-pairs 0 --no-other-workloads"
for SET in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
           "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS" \
           "SQ_INST_LEVEL_EXP SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL"; do
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
for k in sorted(tot): print(f"{k}\t{tot[k]:.6g}")
PY
