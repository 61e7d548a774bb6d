"""The RCCL leg of the multi-GPU path, executed on the GPU box with ONE rank (the box has one card; the driver's
8-GPU run must not be the first time the two libraries meet): a child process initialises torch.distributed with
backend "nccl" (= RCCL on ROCm) BEFORE any other GPU call, then opens the decode library (libfastsmc_hip.so, linked
against the system ROCm next to the runtime torch bundles -- the two-runtimes hazard of DESIGN.md §1), decodes a small
list through the C ABI and sends its records through the path's collectives (all_gather of counts + gather of padded
payloads, device="cuda", `force_collective=True`).  The gathered stream must equal the no-collective path's and the
oracle's.  Also the in-memory product consumer (`gather_hmm_records`) and bench.py's N = 1 `--force-collective` line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))   # RCCL first
t = torch.ones(4, device="cuda")
dist.all_reduce(t)                      # the communicator exists and works before the library is even mapped
assert float(t.sum().item()) == 4.0
from fastsmc_amd import capi, synth     # ... now libfastsmc_hip.so
from fastsmc_amd.dist import gather_ibd_records
from oracle import oracle as O

tables = synth.make_model_tables(69)
haps = synth.make_haps(64, 640, seed=7, cm_per_mb=25.0, switch_per_cm=0.6)
bits, derived, flipped = synth.fold_and_pack(haps.alleles)
folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
gen = (haps.cm / 100.0).astype(np.float32)
pm = O.prepare_model(tables, gen, haps.bp, derived, 64, time=50)
pairs = O.enumerate_all_pairs(32)[:200]
want = O.decode_pairs_ibd(pm, folded, pairs, batch_size=64)

ctx = capi.Context(0)
model = ctx.create_model(pm)
ctx.upload_haps(bits, pm.S)
pr = np.array(pairs, dtype=np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
rec = ctx.decode_ibd(model, pr, capi.whole_sequence_groups(len(pairs), pm.S))
n0, plain = gather_ibd_records(rec, 5, None, 0, 1)                              # no collective
n1, coll = gather_ibd_records(rec, 5, dist, 0, 1, device="cuda", force_collective=True)   # RCCL, one rank
assert n0 == n1 == plain.size == coll.size and plain.size > 0
assert plain.dtype == coll.dtype and np.array_equal(plain, coll)
assert np.array_equal(coll["pair"], want["pair"] + 5)
for f_got, f_want in (("start", "start"), ("end", "end"), ("prob", "prob"), ("post_mean", "postMean"), ("map", "map")):
    assert np.array_equal(coll[f_got], want[f_want]), f_got
# a second decode AFTER the collectives (the communicator's streams and buffers next to the library's)
rec2 = ctx.decode_ibd(model, pr, capi.whole_sequence_groups(len(pairs), pm.S))
assert np.array_equal(rec, rec2)
# an empty rank's payload (no records) goes through the same collectives
n2, empty = gather_ibd_records(rec[:0], 0, dist, 0, 1, device="cuda", force_collective=True)
assert n2 == 0 and empty.size == 0
ctx.close()
dist.barrier()
dist.destroy_process_group()
paths = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64.so" in l or "librccl" in l})
print("RCCL_ONE_RANK_OK", plain.size, json_paths := "|".join(paths))
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    return dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))


def test_rccl_collectives_and_the_decode_library_share_a_process():
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, env=_env(), cwd=ROOT,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_ONE_RANK_OK")]
    assert line, r.stdout[-1500:]
    assert int(line[0].split()[1]) > 0
    assert "librccl" in line[0]  # the collective ran in RCCL, not in a fallback


def test_bench_line_with_the_collective_forced(tmp_path):
    """bench.py at N = 1 with `--force-collective`: the step's record gather goes through RCCL (backend nccl, one
    rank); the gathered stream is the plain run's."""
    shape = ["--workload", "c3", "--haps", "64", "--sites", "700", "--pairs", "1500", "--steps", "1", "--warmup", "1",
             "--cpu-pairs", "0", "--no-other-workloads"]
    outs = []
    for extra, name in (([], "plain.npy"), (["--force-collective"], "rccl.npy")):
        path = str(tmp_path / name)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *shape, *extra,
                            "--dump-records", path], capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
        assert line["n_gpus"] == 1 and line["config"].get("record_gather", "none") == ("rccl" if extra else "none")
        outs.append(np.load(path))
    assert outs[0].size > 0 and np.array_equal(outs[0], outs[1])


def test_in_memory_product_records_through_rccl(tmp_path):
    """`gather_hmm_records` (the product's in-memory consumer) with the collective forced, in a child that brings RCCL
    up first: equals the no-collective records of the same run."""
    child = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from fastsmc_amd import api
from fastsmc_amd import dist as fd
from test_gpu_api import _params, make_files
files = make_files(sys.argv[2])
f = api.FastSMC(_params(files, os.path.join(sys.argv[2], "out")))
f.hmm().setKeepIbdRecords(True)
f.run()
n0, plain = fd.gather_hmm_records(f.hmm())
n1, coll = fd.gather_hmm_records(f.hmm(), dist, 0, 1, device="cuda", force_collective=True)
assert n0 == n1 and n0 > 20 and np.array_equal(plain, coll)
dist.destroy_process_group()
print("HMM_RECORDS_OK", n0)
"""
    r = subprocess.run([sys.executable, "-c", child, ROOT, str(tmp_path)], capture_output=True, text=True, env=_env(),
                       cwd=ROOT, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "HMM_RECORDS_OK" in r.stdout


def test_posterior_sums_through_rccl(tmp_path):
    """Sum-over-pairs mode: the merge of the per-rank planes (`fastsmc_amd.dist.reduce_sums`, a gather to rank 0 and the
    rank-ordered fp32 addition) with the collective forced through RCCL on one rank -- `0 + P_0` is `P_0`: the plane of
    the plain run, bit for bit -- and the re-associating all-reduce variant on the same communicator."""
    shape = ["--mode", "sums", "--workload", "c1", "--haps", "64", "--sites", "400", "--steps", "1", "--warmup", "1",
             "--cpu-pairs", "0"]
    outs = []
    for extra, name in (([], "plain.npy"), (["--force-collective"], "rccl.npy")):
        path = str(tmp_path / name)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *shape, *extra,
                            "--dump-records", path], capture_output=True, text=True, env=_env(), cwd=ROOT, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1])
        assert line["config"]["mode"] == "sums"
        assert line["config"]["sums_reduction"] == ("rank order on rank 0" if extra else "none")
        outs.append(np.load(path))
    assert outs[0].shape == (400, 69) and outs[0].any() and np.array_equal(outs[0], outs[1])
    child = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from fastsmc_amd.dist import reduce_sums
rng = np.random.default_rng(3)
planes = {n: rng.random((50, 12), dtype=np.float32) for n in ("sumOverPairs", "sumOverPairs00", "sumOverPairs01", "sumOverPairs11")}
for order in ("rank", "allreduce"):
    got = reduce_sums(planes, dist, 0, 1, device="cuda", order=order, force_collective=True)
    assert list(got) == list(planes) and all(np.array_equal(got[n], planes[n]) for n in planes), order
dist.destroy_process_group()
print("SUMS_RCCL_OK", "|".join(sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})))
"""
    r = subprocess.run([sys.executable, "-c", child, ROOT], capture_output=True, text=True, env=_env(), cwd=ROOT,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "SUMS_RCCL_OK" in r.stdout and "librccl" in r.stdout


def test_product_driver_gathers_its_records_through_rccl(tmp_path):
    """`run_fastsmc_sharded` on an initialised "nccl" group (one rank, collective forced): no part file, the records
    travel through RCCL from `cuda:0` and rank 0 writes the file -- the bytes of a plain FastSMC.run()."""
    child = r"""
import gzip, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from fastsmc_amd import api
from fastsmc_amd import dist as fd
from test_gpu_api import _params, make_files
files = make_files(sys.argv[2])
for kw in (dict(), dict(BIN_OUT=True), dict(hashing=True, min_m=1.0)):
    f = api.FastSMC(_params(files, os.path.join(sys.argv[2], "plain"), **kw))
    f.run()
    want = gzip.open(f.outputFileName(), "rb").read()
    out = fd.run_fastsmc_sharded(_params(files, os.path.join(sys.argv[2], "rccl"), **kw), rank=0, world=1, local_rank=0,
                                 gather="records", force_collective=True)
    assert len(want) > 1000 and gzip.open(out, "rb").read() == want, kw
dist.destroy_process_group()
print("DRIVER_RCCL_OK", "|".join(sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})))
"""
    r = subprocess.run([sys.executable, "-c", child, ROOT, str(tmp_path)], capture_output=True, text=True, env=_env(),
                       cwd=ROOT, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "DRIVER_RCCL_OK" in r.stdout and "librccl" in r.stdout
