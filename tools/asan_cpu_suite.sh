#!/bin/bash
# CPU-only sanitizer pass (the reference's CI has an ASan job; GPU sanitizers are not available on the pool): builds the
# oracle and the host module (fastsmc_amd/csrc/host/*.cpp -> _pyasmc) with -fsanitize=address,undefined, runs the CPU
# test suite with libasan / libubsan preloaded into python, then puts the regular builds back.
# Usage: bash tools/asan_cpu_suite.sh        (from the repository root; about three minutes)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
MOD=$(python -c "from fastsmc_amd import build as b; print(b.host_module_path())")
PYINC=$(python -c "import sysconfig; print(sysconfig.get_paths()['include'])")
PBINC=$(python -c "import pybind11; print(pybind11.get_include())")
TMP=$(mktemp -d)
cp $MOD $TMP/pyasmc.bak
cp oracle/liboracle.so $TMP/liboracle.bak
restore() { cp $TMP/pyasmc.bak $MOD; cp $TMP/liboracle.bak oracle/liboracle.so; rm -rf $TMP; }
trap restore EXIT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g -fPIC -ffp-contract=off"
gcc $SAN -std=c99 -c oracle/hmm_oracle.c -o $TMP/hmm_oracle.o
g++ $SAN -std=c++17 -c oracle/undist_counts.cpp -o $TMP/undist.o
g++ -shared -fsanitize=address,undefined -o oracle/liboracle.so $TMP/hmm_oracle.o $TMP/undist.o -lm
PIDS=""
for f in decoding_quantities decoding_params data hmm hashing drivers pybind_module pybind_containers; do
  g++ $SAN -std=c++17 -fvisibility=hidden -I $PBINC -I $PYINC -c fastsmc_amd/csrc/host/$f.cpp -o $TMP/$f.o & PIDS="$PIDS $!"
done
for P in $PIDS; do wait $P; done
g++ -shared -fsanitize=address,undefined -o $MOD $TMP/decoding_quantities.o $TMP/decoding_params.o $TMP/data.o $TMP/hmm.o \
    $TMP/hashing.o $TMP/drivers.o $TMP/pybind_module.o $TMP/pybind_containers.o -L fastsmc_amd -lfastsmc_hip -lz -Wl,-rpath,'$ORIGIN'
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  python -m pytest tests -x -q -m "not gpu" -k "not isa_hazards and not multiproc"
