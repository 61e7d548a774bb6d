"""bench.py's multi-GPU mode (strong scaling of ONE sharded work list): a real 2-process run -- two ranks on the one
card of the GPU box, gloo as the control plane (RCCL wants one device per rank) -- returns the same ordered record
stream as the 1-process run of the same list, and both print the contract's JSON line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = ["--workload", "c3", "--haps", "64", "--sites", "700", "--pairs", "1500", "--steps", "1", "--warmup", "1",
         "--cpu-pairs", "0"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_return_the_single_process_record_stream(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = str(tmp_path / "one.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *SHAPE, "--dump-records", one],
                        capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    line1 = json.loads(r1.stdout.strip().splitlines()[-1])
    two = str(tmp_path / "two.npy")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                         os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", *SHAPE,
                         "--dump-records", two],
                        capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    line2 = json.loads([ln for ln in r2.stdout.strip().splitlines() if ln.startswith("{")][-1])
    a, b = np.load(one), np.load(two)
    assert a.size > 0 and a.dtype == b.dtype and np.array_equal(a, b)
    assert np.all(np.diff(a["pair"].astype(np.int64)) >= 0)
    for line, n in ((line1, 1), (line2, 2)):
        assert line["n_gpus"] == n and line["scaling"] == "strong" and line["unit"] == "pairs/s"
        assert line["config"]["ibd_records_per_step"] == a.size
        assert "sharded over" in line["config"]["workload"] and line["roofline"]["bound"] == "hbm"
    assert len(line2["config"]["kernel_ms_per_rank"]) == 2


def test_sum_over_pairs_of_two_ranks_is_the_rank_ordered_merge(tmp_path):
    """`bench.py --mode sums`: rank r decodes job r + 1 of 2 (HMM.cpp:319-321) and `reduce_sums` merges the planes on
    rank 0 in rank order.  Against ONE process that decodes all pairs the merged plane differs by fp32 re-association at
    the one shard join only (1e-6 relative); against the two jobs' planes, decoded one after the other by one process
    and added in job order, it is the same bits."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    shape = ["--mode", "sums", "--workload", "c1", "--haps", "64", "--sites", "400", "--steps", "1", "--warmup", "1",
             "--cpu-pairs", "0"]
    one = str(tmp_path / "one.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *shape, "--dump-records", one],
                        capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    two = str(tmp_path / "two.npy")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                         os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", *shape,
                         "--dump-records", two],
                        capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    line2 = json.loads([ln for ln in r2.stdout.strip().splitlines() if ln.startswith("{")][-1])
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape == (400, 69) and a.any()
    np.testing.assert_allclose(b, a, rtol=1e-6, atol=1e-7)
    assert line2["n_gpus"] == 2 and line2["scaling"] == "strong" and line2["config"]["mode"] == "sums"
    assert line2["config"]["sums_reduction"] == "rank order on rank 0"
    # the same two jobs by one process, added in job order: bit for bit
    sys.path.insert(0, ROOT)
    import bench
    from fastsmc_amd import capi

    pm, bits, _, _ = bench.build_problem(64, 400, 69, seed=1234)
    every = bench.all_pairs(32)
    ctx = capi.Context(0)
    model = ctx.create_model(pm)
    ctx.upload_haps(bits, pm.S)
    acc = np.zeros((pm.S, pm.K), np.float32)
    for r in range(2):
        lo, hi = every.shape[0] * r // 2, every.shape[0] * (r + 1) // 2
        ctx.upload_worklist(every[lo:hi].view(capi.PAIR_DTYPE).reshape(-1), capi.whole_sequence_groups(hi - lo, pm.S, batch=64))
        plane, _ = ctx.decode_sums(model)
        acc = acc + plane
    ctx.close()
    np.testing.assert_array_equal(b, acc)
