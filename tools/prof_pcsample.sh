#!/bin/bash
# Runs on the GPU box: PC sampling of a reduced bench workload (rocprofv3 beta feature), to see where waves sit.
# Usage: tools/prof_pcsample.sh <tag> <method: stochastic|host_trap> [bench args...]
set -u
TAG=$1; METHOD=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pcs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
if [ "$METHOD" = stochastic ]; then UNIT=cycles; INT=${PCS_INTERVAL:-65536}; else UNIT=time; INT=${PCS_INTERVAL:-100}; fi
rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INT \
  --output-format csv -d $OUT/raw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-pairs 0 "$@" > $OUT/run.log 2>&1
echo "rc=$?" >> $OUT/run.log
ls -R $OUT/raw | head -20 >> $OUT/run.log
for f in $(find $OUT/raw -name "*pc_sampling*csv"); do
  wc -l $f >> $OUT/run.log
  head -3 $f >> $OUT/run.log
  # keep the sample file small: gzip it
  gzip -f $f
done
tail -30 $OUT/run.log
