#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 passes over the default bench.py command.
#   1. --kernel-trace --stats           -> per-kernel durations (must agree with bench.py's hipEvent timing)
#   2. --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing)  -> HBM traffic per launch
# Usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (the headline kernel alone: the default line's other workloads launch kernels of the same name)
ARGS="$GRAFT_REPO_ROOT/bench.py --no-other-workloads $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_under_trace.json 2> $OUT/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 $ARGS --cpu-pairs 0 > $OUT/bench_under_pmc_$C.json 2> $OUT/pmc_$C.err
done
python3 - <<PY
import csv, glob, json
out="$OUT"
res={}
for f in glob.glob(out+"/trace/**/*kernel_stats.csv", recursive=True):
    res["kernel_stats_csv"]=open(f).read()
per={}
for f in glob.glob(out+"/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "decode_kernel" in r.get("Kernel_Name",""):
            key=(r["Counter_Name"], r["Dispatch_Id"])
            per[key]=per.get(key,0.0)+float(r["Counter_Value"])
agg={}
for (c,d),v in per.items():
    agg.setdefault(c,[]).append(v)
res["pmc_per_launch"]={c: {"launches": len(v), "mean": sum(v)/len(v), "values": v} for c,v in agg.items()}
json.dump(res, open(out+"/summary.json","w"), indent=1)
print(res.get("kernel_stats_csv","")); print(json.dumps(res["pmc_per_launch"]))
PY
