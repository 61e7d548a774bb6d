#!/bin/bash
# ThreadSanitizer pass over the multi-threaded host start-up (Data::readHapsFastSMC: reader thread + parser pool + parallel
# transpose; Data::calculateUndistinguishedCounts: parallel shuffles): a standalone C++ harness -- python cannot preload
# libtsan -- reads a small synthetic .hap.gz cut into many blocks with six threads and job windows.  No GPU.
# Usage: bash tools/tsan_host_reader.sh      (prints the harness' one line; any data race is a TSan report on stderr)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
trap 'rm -rf $TMP' EXIT
cd $ROOT
python - <<PY
import sys
sys.path.insert(0, "$ROOT")
from fastsmc_amd import synth
synth.write_haps_files_fast("$TMP/t", synth.make_haps_blocked(400, 3000, seed=5), block=64)
PY
cat > $TMP/main.cpp <<'CPP'
#include <cstdio>
#include <cstdlib>
#include "data.hpp"
#include "decoding_params.hpp"
using namespace fsmc_host;
int main(int, char** argv)
{
  setenv("FSMC_HOST_THREADS", "6", 1);
  setenv("FSMC_HOST_BLOCK_BYTES", "20000", 1);
  DecodingParams p;
  p.inFileRoot = argv[1];
  p.FastSMC = true;
  p.foldData = true;
  p.useKnownSeed = true;
  p.jobs = 4;
  p.jobInd = 2;
  Data d(p);
  const auto u = d.calculateUndistinguishedCounts(50);
  unsigned long long ones = 0;
  for (auto w : d.bits) ones += __builtin_popcountll(w);
  std::printf("sites %d haplotype rows %zu ones %llu undistinguished[7][1] %d\n", d.sites, d.numHapRows(), ones, u[7][1]);
  return 0;
}
CPP
g++ -std=c++17 -O1 -g -fsanitize=thread -I fastsmc_amd/csrc/host -I include $TMP/main.cpp fastsmc_amd/csrc/host/data.cpp \
    fastsmc_amd/csrc/host/decoding_params.cpp -lz -lpthread -o $TMP/tsan_reader
$TMP/tsan_reader $TMP/t
