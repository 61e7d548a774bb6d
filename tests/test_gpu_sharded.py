"""Multi-GPU product path (scope row e): one job sharded over `world` processes, each decoding a contiguous range
of whole batches; the concatenated part files must decompress to exactly the single-device output -- text and
binary, all-pairs and hashing mode.  The GPU box has one device, so all ranks use device 0."""
import gzip
import multiprocessing as mp
import os
import socket

import pytest

from fastsmc_amd import api, dist
from test_gpu_api import _params, files  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _single(files, out, **kw):
    p = _params(files, out, **kw)
    f = api.FastSMC(p)
    f.run()
    return gzip.open(f.outputFileName(), "rb").read()


@pytest.mark.parametrize("kw", [dict(), dict(BIN_OUT=True), dict(hashing=True, min_m=1.0),
                                dict(jobs=9, jobInd=7)], ids=["text", "binary", "hashing", "job7of9"])
def test_sharded_in_process_equals_single_device(files, tmp_path, kw):
    want = _single(files, str(tmp_path / "one"), **kw)
    assert len(want) > 1000
    world = 3
    final = None
    for rank in (1, 2, 0):  # rank 0 last: it concatenates the parts
        p = _params(files, str(tmp_path / "sharded"), **kw)
        r = dist.run_fastsmc_sharded(p, rank=rank, world=world, local_rank=0, barrier=lambda: None)
        final = r or final
    assert final and not final.endswith(f"of{world}")
    assert gzip.open(final, "rb").read() == want
    assert not [n for n in os.listdir(tmp_path) if ".part" in n]


def _worker(rank, world, port, files, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    p = _params(files, out, hashing=True, min_m=1.0)
    dist.run_fastsmc_sharded(p, rank=rank, world=world, local_rank=0)


def test_sharded_two_processes(files, tmp_path):
    want = _single(files, str(tmp_path / "one"), hashing=True, min_m=1.0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = str(tmp_path / "two")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, files, out)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(600)
        assert pr.exitcode == 0
    assert gzip.open(out + ".1.1.FastSMC.ibd.gz", "rb").read() == want


@pytest.mark.parametrize("kw", [dict(), dict(hashing=True, min_m=1.0)], ids=["all-pairs", "hashing"])
def test_in_memory_records_of_the_shards_are_the_single_device_stream(files, tmp_path, kw):
    """The in-memory consumer (HMM.getIbdRecordArrays + dist.gather_hmm_records, the RCCL gather of the product): the
    shards' kept records, in rank order, are the records of the one-device run -- every field, every bit."""
    import numpy as np

    def records(rank, world):
        f = api.FastSMC(_params(files, str(tmp_path / f"m{rank}of{world}"), **kw))
        f.setShard(rank, world)
        f.hmm().setKeepIbdRecords(True)
        f.run()
        return dist.gather_hmm_records(f.hmm())[1]

    one = records(0, 1)
    assert one.size > 20
    parts = [records(r, 3) for r in range(3)]
    merged = np.concatenate(parts)
    for name in one.dtype.names:
        if name != "pair":  # (ordinals count from the start of each rank's shard)
            np.testing.assert_array_equal(merged[name], one[name], err_msg=name)
    assert all((np.diff(p["pair"].astype(np.int64)) >= 0).all() for p in parts)


def _worker_kw(rank, world, port, files, out, kw, gather):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.run_fastsmc_sharded(_params(files, out, **kw), rank=rank, world=world, local_rank=0, gather=gather)


@pytest.mark.parametrize("gather", ["records", "files"])
@pytest.mark.parametrize("kw", [dict(BIN_OUT=True), dict()], ids=["binary", "text"])
def test_sharded_two_processes_both_routes(files, tmp_path, kw, gather):
    """Two processes on the one card with a gloo group: the records gathered over the process group and written by rank
    0 (`gather="records"`, the route an RCCL group takes too) and the part files (`gather="files"`) both decompress to
    the single-device output, text and binary (header once)."""
    want = _single(files, str(tmp_path / "one"), **kw)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = str(tmp_path / "two")
    procs = [ctx.Process(target=_worker_kw, args=(r, 2, port, files, out, kw, gather)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(600)
        assert pr.exitcode == 0
    name = out + (".1.1.FastSMC.bibd.gz" if kw.get("BIN_OUT") else ".1.1.FastSMC.ibd.gz")
    assert gzip.open(name, "rb").read() == want
    assert not [n for n in os.listdir(tmp_path) if ".part" in n]
