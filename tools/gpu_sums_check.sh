#!/bin/bash
# Runs on the GPU box: the sums consumer with beta stride 2 and resident chunks -- parity tests, then C2-size and C1-size
# timings with stride 1 and 2.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/sums_check; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_gpu_beta_stride.py tests/test_gpu_modes.py tests/test_gpu_sums_order.py tests/test_gpu_two_wave_windows.py tests/test_gpu_resident_chunks.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python3 - <<'PY'
import sys, json, time
sys.path.insert(0, ".")
import numpy as np
import bench
from fastsmc_amd import capi
for name, (n_hap, n_sites) in (("c2", (1000, 50000)), ("c1", (300, 6760))):
    pm, bits, _, _ = bench.build_problem(n_hap, n_sites, 69, seed=1234)
    pairs = bench.all_pairs(n_hap // 2)
    for stride in (1, 2, 1, 2):
        ctx = capi.Context(0)
        ctx.set_workspace_limit(int(0.8 * ctx.info()["hbm_bytes"]))
        ctx.set_beta_stride(stride)
        ctx.set_two_wave_windows(1)
        model = ctx.create_model(pm)
        ctx.upload_haps(bits, pm.S)
        ctx.upload_worklist(pairs.view(capi.PAIR_DTYPE).reshape(-1), capi.whole_sequence_groups(int(pairs.shape[0]), pm.S, batch=64))
        ctx.decode_sums(model)
        s, _ = ctx.decode_sums(model)
        ms = ctx.last_kernel_ms()
        algo = bench.algorithmic_bytes(int(pairs.shape[0]), pm.S, pm.K)
        print(name, "stride", stride, "kernel_ms %.1f" % ms, "frac %.3f" % (algo / (ms / 1e3) / 8e12), "resident", ctx.last_resident_chunks(), "chunks", ctx.info()["max_chunks"], "checksum %.6e" % float(s.astype(np.float64).sum()), flush=True)
        ctx.close()
PY
