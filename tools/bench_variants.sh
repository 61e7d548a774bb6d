#!/bin/bash
# Runs on the GPU box: short C2 bench of the shipped library and of every variant named (fastsmc_amd/variants/lib<name>.so).
# Usage: tools/bench_variants.sh <tag> <name> [<name> ...]   (name "base" = the shipped library)
TAG=$1; shift
mkdir -p gpurun_out
for V in "$@"; do
  if [ "$V" = base ]; then unset FSMC_HIP_LIB; else export FSMC_HIP_LIB=$PWD/fastsmc_amd/variants/lib$V.so; fi
  python bench.py --steps 2 --warmup 1 --cpu-pairs 0 ${BENCH_ARGS:-} > gpurun_out/${TAG}_$V.json 2> gpurun_out/${TAG}_$V.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${TAG}_$V.json"))
    print("$V", "kernel_ms", round(d["roofline"]["kernel_ms"],1), "pairs/s", round(d["value"]), "records", d["config"]["ibd_records_per_step"], d["config"].get("phase_cycles",""))
except Exception as e:
    print("$V", "FAILED", e)
PY
done
