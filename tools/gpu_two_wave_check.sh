#!/bin/bash
# Runs on the GPU box: the two-waves-per-window kernel -- parity tests, then the C1-shape consumers with and without it,
# then the product path's timeline on C2 files (FSMC_HOST_TIMING).
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/two_wave_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_two_wave_windows.py tests/test_gpu_modes.py tests/test_gpu_sums_order.py tests/test_gpu_generic_k.py tests/test_gpu_per_pair_files.py tests/test_gpu_resident_chunks.py tests/test_gpu_sharded.py tests/test_gpu_rccl_one_rank.py tests/test_gpu_bench_strong.py -x -q -m gpu > $OUT/pytest.log 2>&1
RC=$?
echo "pytest rc=$RC"; tail -5 $OUT/pytest.log
if [ $RC -ne 0 ]; then exit $RC; fi
timeout -k 10 300 python3 tools/measure_configs.py c1_consumers > $OUT/c1_consumers.jsonl 2> $OUT/c1_consumers.err
echo "c1_consumers rc=$?"; cat $OUT/c1_consumers.jsonl
FSMC_HOST_TIMING=1 timeout -k 10 600 python3 tools/measure_configs.py run_c2 > $OUT/run_c2.json 2> $OUT/run_c2.err
echo "run_c2 rc=$?"; cat $OUT/run_c2.json; grep "fsmc" $OUT/run_c2.err
