#!/bin/bash
# Runs on the GPU box: same-box A/B of library builds.  Usage: tools/r03_ab.sh <tag> "<bench args>" <name> [<name>...]
# (name "base" = the shipped library, anything else fastsmc_amd/variants/lib<name>.so); every build is run twice, interleaved.
TAG=$1; shift
export BENCH_ARGS="$1"; shift
cd $GRAFT_REPO_ROOT
for ROUND in 1 2; do
  bash tools/bench_variants.sh ${TAG}_r$ROUND "$@"
done
