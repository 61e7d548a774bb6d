#!/bin/bash
# Builds an experimental variant of the whole HIP library into fastsmc_amd/variants/lib<name>.so (every member the build
# ships, from fastsmc_amd/build.py's lists): tools/build_variant.py does the work.
# Usage: tools/build_variant.sh <name> [extra hipcc flags, e.g. -DFSMC_PHASE_STAMPS]
set -eu
NAME=$1; shift
exec python3 "$(dirname "$0")/build_variant.py" "$NAME" --only all -- "$@"
