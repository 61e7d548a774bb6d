#!/bin/bash
# Builds an experimental variant of the wave-group kernel only (fsmc_kernels_w2.h: 128 < K <= 256) into
# fastsmc_amd/variants/lib<name>.so: the two w2 members are recompiled with the extra flags, every other object is the
# shipped build's (fastsmc_amd/csrc/obj/, so build the library first).  The in-flight scalar-load check
# (tools/check_inflight_sgprs.py) runs on the variant's ISA and the kernel's registers / spills / LDS are printed.
# Usage: tools/build_w2_variant.sh <name> [extra hipcc flags, e.g. -DFSMC_W2_X=1]
#        FSMC_HIP_LIB=fastsmc_amd/variants/lib<name>.so python bench.py --workload c4 ...
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/fastsmc_amd/variants
OBJ=$OUT/obj_$NAME
mkdir -p $OBJ
FLAGS="-std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wno-pass-failed"
SRC=${FSMC_SRC_DIR:-$ROOT/fastsmc_amd/csrc}/fsmc_inst.hip  # (FSMC_SRC_DIR: an edited COPY of csrc/ for diagnostic builds -- the library's own sources, hence its hash, stay as they are)
PIDS=""
for K in 48 64; do
  hipcc $FLAGS "$@" -c -DFSMC_INSTANCE_W2=$K -Rpass-analysis=kernel-resource-usage -o $OBJ/w2_$K.o $SRC 2> $OBJ/w2_$K.res & PIDS="$PIDS $!"
  hipcc $FLAGS "$@" -DFSMC_INSTANCE_W2=$K -S --cuda-device-only -Wno-unused-command-line-argument -o $OBJ/w2_$K.s $SRC 2>/dev/null & PIDS="$PIDS $!"
done
# FSMC_VARIANT_CAPI=1: the variant changes the host side of the launch as well -- fsmc_capi.hip of FSMC_SRC_DIR is compiled too
CAPI=$ROOT/fastsmc_amd/csrc/obj/capi.o
if [ "${FSMC_VARIANT_CAPI:-0}" = "1" ]; then
  CAPI=$OBJ/capi.o
  hipcc $FLAGS "$@" -c -o $CAPI $(dirname $SRC)/fsmc_capi.hip & PIDS="$PIDS $!"  # (default exact members)
fi
for P in $PIDS; do wait $P; done
for K in 48 64; do
  python3 - "$OBJ/w2_$K.s" <<'PY'
import sys
sys.path.insert(0, "tools")
import check_inflight_sgprs as chk
rc = chk.check(sys.argv[1])
print("in-flight SGPR check", sys.argv[1].split("/")[-1], "OK" if rc == 0 else "VIOLATION")
sys.exit(rc)
PY
done
# registers / spills / scratch / LDS of the array-mode IBD kernel with segment ages (the C4 kernel)
python3 - "$OBJ/w2_64.res" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
blocks = txt.split("Function Name: ")
for b in blocks[1:]:
    name = b.split()[0]
    if "Li64ELi0ELb1ELb0E" not in name:
        continue
    f = lambda key: (re.search(key + r": (\d+)", b) or [None, "?"])[1]
    print("w2<64, ibd, track>:", "VGPRs", f("VGPRs"), "AGPRs", f("AGPRs"), "spill VGPR", f("VGPRs Spill"), "spill SGPR", f("SGPRs Spill"),
          "scratch", f("ScratchSize \[bytes/lane\]"), "LDS", f("LDS Size \[bytes/block\]"), "occupancy", f("Occupancy \[waves/SIMD\]"))
PY
OBJS="$(ls $ROOT/fastsmc_amd/csrc/obj/*.o | grep -v "/w2_48.o\|/w2_64.o\|/capi.o") $CAPI"  # (the 80 ... 112-state members are the shipped build's)
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib$NAME.so $OBJS $OBJ/w2_48.o $OBJ/w2_64.o
rm -rf $OBJ
echo "built $OUT/lib$NAME.so"
