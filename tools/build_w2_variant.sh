#!/bin/bash
# Builds an experimental variant of the wave-group kernel only (fsmc_kernels_w2.h: EVERY member of the build's W2_MEMBERS
# list, 129 ... 512 states) into fastsmc_amd/variants/lib<name>.so; every other object is the shipped build's.
# FSMC_SRC_DIR: an edited copy of csrc/; FSMC_VARIANT_CAPI=1: fsmc_capi.hip is recompiled as well.
# Usage: tools/build_w2_variant.sh <name> [extra hipcc flags, e.g. -DFSMC_W2_X=1]
#        FSMC_HIP_LIB=fastsmc_amd/variants/lib<name>.so python bench.py --workload c4 ...
set -eu
NAME=$1; shift
ARGS="--only w2 --resources decode_kernel_w2ILi64ELi0ELb1ELb0E"
if [ -n "${FSMC_SRC_DIR:-}" ]; then ARGS="$ARGS --src-dir $FSMC_SRC_DIR"; fi
if [ "${FSMC_VARIANT_CAPI:-0}" = "1" ]; then ARGS="$ARGS --capi"; fi
exec python3 "$(dirname "$0")/build_variant.py" "$NAME" $ARGS -- "$@"
