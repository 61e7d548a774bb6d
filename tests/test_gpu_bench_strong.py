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
