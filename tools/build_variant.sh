#!/bin/bash
# Builds an experimental variant of the HIP library (same C ABI) into fastsmc_amd/variants/lib<name>.so;
# run it with FSMC_HIP_LIB=fastsmc_amd/variants/lib<name>.so python bench.py ...
# Only the members a C2-style bench needs are built: fsmc_capi.hip + the K = 69 member (+ the others as stubs is not
# possible: the selection code references them), so every member is compiled -- in parallel.
# Usage: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DFSMC_PHASE_STAMPS]
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/fastsmc_amd/variants
OBJ=$OUT/obj_$NAME
mkdir -p $OBJ
FLAGS="-std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-pass-failed -c"
PIDS=""
hipcc $FLAGS "$@" -o $OBJ/capi.o $ROOT/fastsmc_amd/csrc/fsmc_capi.hip & PIDS="$PIDS $!"
for K in 16 32 48 50 64 69 80 96 100 112 128; do
  hipcc $FLAGS "$@" -DFSMC_INSTANCE_KT=$K -o $OBJ/kt$K.o $ROOT/fastsmc_amd/csrc/fsmc_inst.hip & PIDS="$PIDS $!"
done
for K in 48 64 80; do
  hipcc $FLAGS "$@" -DFSMC_INSTANCE_W2=$K -o $OBJ/w2_$K.o $ROOT/fastsmc_amd/csrc/fsmc_inst.hip & PIDS="$PIDS $!"
done
for N in 6 7 8; do
  hipcc $FLAGS "$@" -DFSMC_INSTANCE_W2=64 -DFSMC_INSTANCE_NW=$N -o $OBJ/w2_64x$N.o $ROOT/fastsmc_amd/csrc/fsmc_inst.hip & PIDS="$PIDS $!"
done
hipcc $FLAGS -o $OBJ/idsort.o $ROOT/fastsmc_amd/csrc/fsmc_identify_sort.hip & PIDS="$PIDS $!"
hipcc $FLAGS -o $OBJ/idseeds.o $ROOT/fastsmc_amd/csrc/fsmc_identify_seeds.hip & PIDS="$PIDS $!"
for P in $PIDS; do wait $P; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$NAME.so $OBJ/*.o
rm -rf $OBJ
