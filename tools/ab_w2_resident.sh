#!/bin/bash
# Runs on the GPU box: the wave-group kernel with and without resident chunks, interleaved on one box -- parity tests first,
# then config 4 at size and the C2 shape at 192 states (bench.py --resident-chunks 0 | -1).
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/w2res_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_resident_chunks.py tests/test_gpu_generic_k.py tests/test_gpu_wide_model.py tests/test_gpu_sequence.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_modes.py -x -q -m gpu > $OUT/pytest.log 2>&1
RC=$?
echo "pytest rc=$RC"; tail -8 $OUT/pytest.log | cut -c1-300
if [ $RC -ne 0 ]; then exit $RC; fi
show() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', 'resident', d['config']['resident_chunks'], 'chunks', d['config']['chunks_per_window'], 'kernel_ms %.1f frac %.4f records %d' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['config']['ibd_records_per_step']))"; }
for rep in 1 2; do
  for R in 0 -1; do
    timeout -k 10 300 python3 bench.py --workload c4 --steps 1 --warmup 1 --cpu-pairs 0 --resident-chunks $R > $OUT/c4_$R.json 2> $OUT/c4_$R.err && show $OUT/c4_$R.json "c4 R=$R"
  done
done
for R in 0 -1; do
  timeout -k 10 300 python3 bench.py --states 192 --steps 1 --warmup 1 --cpu-pairs 0 --no-other-workloads --resident-chunks $R > $OUT/k192_$R.json 2> $OUT/k192_$R.err && show $OUT/k192_$R.json "k192 c2-shape R=$R"
done
