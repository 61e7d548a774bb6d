#!/bin/bash
# Builds an experimental variant of the HIP library (same C ABI) into fastsmc_amd/variants/lib<name>.so;
# run it with FSMC_HIP_LIB=fastsmc_amd/variants/lib<name>.so python bench.py ...
# Usage: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DFSMC_PHASE_STAMPS]
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/fastsmc_amd/variants
hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -Wno-pass-failed \
  "$@" -o $ROOT/fastsmc_amd/variants/lib$NAME.so $ROOT/fastsmc_amd/csrc/fsmc_capi.hip
