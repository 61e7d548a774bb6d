"""ctypes binding of the C ABI in ``include/fastsmc_hip.h`` (``libfastsmc_hip.so``).

This is plumbing, not the product: it lets Python tests, ``bench.py`` and the Python-level
drivers call the HIP library through exactly the symbols a C/C++ host would bind.  There is no
fallback of any kind here -- if the shared library is missing or no MI355X is visible, calls
raise ``FsmcError``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FSMC_HIP_LIB lets a profiling run point at an experimental build of the same ABI
LIB_PATH = os.environ.get("FSMC_HIP_LIB") or os.path.join(_HERE, "libfastsmc_hip.so")

FSMC_WANT_MEAN = 1
FSMC_WANT_MAP = 2
FSMC_WANT_SUMS = 4
FSMC_WANT_MAJOR_MINOR_SUMS = 8

FSMC_EOVERFLOW = -6

# every symbol include/fastsmc_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "fsmc_ctx_create", "fsmc_ctx_destroy", "fsmc_last_error", "fsmc_ctx_info", "fsmc_ctx_set_workspace_limit",
    "fsmc_ctx_expect_work",
    "fsmc_ctx_set_chunk_sites", "fsmc_ctx_set_beta_stride", "fsmc_ctx_last_beta_stride", "fsmc_ctx_last_plan",
    "fsmc_ctx_last_kernel", "fsmc_ctx_set_pairing", "fsmc_ctx_last_items", "fsmc_ctx_set_two_wave_windows", "fsmc_ctx_last_waves_per_window", "fsmc_ctx_last_segment_sums_in_lds", "fsmc_ctx_set_resident_chunks",
    "fsmc_ctx_last_resident_chunks",
    "fsmc_model_create", "fsmc_model_destroy", "fsmc_haps_upload", "fsmc_worklist_upload",
    "fsmc_decode_ibd_launch", "fsmc_decode_ibd_fetch", "fsmc_sync", "fsmc_last_kernel_ms", "fsmc_phase_cycles",
    "fsmc_decode_ibd",
    "fsmc_decode_posteriors", "fsmc_decode_per_pair", "fsmc_decode_sums", "fsmc_decode_sums_batches",
    "fsmc_identify", "fsmc_identify_ex", "fsmc_identify_fetch",
]

PAIR_DTYPE = np.dtype([("hap_a", "<u4"), ("hap_b", "<u4")])
GROUP_DTYPE = np.dtype([("first_pair", "<u4"), ("n_pairs", "<u4"), ("from", "<u4"), ("to", "<u4"),
                        ("scan_from", "<u4"), ("scan_to", "<u4")])
CANDIDATE_DTYPE = np.dtype([("hap_a", "<u4"), ("hap_b", "<u4"), ("from", "<u4"), ("to", "<u4"), ("flush_word", "<u4")])
IBD_DTYPE = np.dtype([("pair", "<u4"), ("start", "<i4"), ("end", "<i4"), ("prob", "<f4"), ("post_mean", "<f4"),
                      ("map", "<f4")])


class FsmcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"fastsmc_hip error {code}: {msg}")
        self.code = code


class _JobWindow(C.Structure):
    _fields_ = [("window_size", C.c_uint32), ("w_i", C.c_uint32), ("w_j", C.c_uint32), ("last_job", C.c_int32),
                ("j_above_diag", C.c_int32)]


class _IdentifyOpts(C.Structure):
    _fields_ = [("word_size", C.c_uint32), ("haploid", C.c_uint32), ("max_seeds", C.c_int32),
                ("read_ahead", C.c_uint32)]


class _ModelDesc(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("S", C.c_int32),
        ("pi", C.c_void_p), ("col_ratios", C.c_void_p), ("exp_times", C.c_void_p),
        ("n_rows", C.c_int32),
        ("D", C.c_void_p), ("B", C.c_void_p), ("U", C.c_void_p), ("RR", C.c_void_p),
        ("step_row", C.c_void_p),
        ("e1", C.c_void_p), ("e0m1", C.c_void_p), ("e2m0", C.c_void_p),
        ("state_threshold", C.c_uint32), ("age_threshold", C.c_uint32), ("probability_threshold", C.c_float),
        ("sequence", C.c_int32),
        ("gap_row_f", C.c_void_p), ("site_row_f", C.c_void_p), ("gap_row_b", C.c_void_p), ("site_row_b", C.c_void_p),
        ("hom", C.c_void_p),
    ]


_lib = None


def load():
    """dlopen the HIP library; raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FsmcError(-100, f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                                  "g.build()'` (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        vp, i32, u32, u64, sz = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_size_t
        L.fsmc_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
        L.fsmc_ctx_destroy.argtypes = [vp]
        L.fsmc_ctx_destroy.restype = None
        L.fsmc_last_error.argtypes = [vp]
        L.fsmc_last_error.restype = C.c_char_p
        L.fsmc_ctx_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(u64)]
        L.fsmc_ctx_set_workspace_limit.argtypes = [vp, u64]
        L.fsmc_ctx_expect_work.argtypes = [vp, C.c_double, C.c_int32]
        L.fsmc_ctx_set_chunk_sites.argtypes = [vp, u32]
        L.fsmc_ctx_set_beta_stride.argtypes = [vp, u32]
        L.fsmc_ctx_last_beta_stride.argtypes = [vp, C.POINTER(i32)]
        L.fsmc_ctx_last_plan.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.fsmc_ctx_last_kernel.argtypes = [vp, C.POINTER(i32)]
        L.fsmc_ctx_set_pairing.argtypes = [vp, u32]
        L.fsmc_ctx_set_two_wave_windows.argtypes = [vp, u32]
        L.fsmc_ctx_last_waves_per_window.argtypes = [vp, C.POINTER(i32)]
        L.fsmc_ctx_last_items.argtypes = [vp, C.POINTER(i32)]
        L.fsmc_ctx_last_segment_sums_in_lds.argtypes = [vp, C.POINTER(i32)]
        L.fsmc_model_create.argtypes = [vp, C.POINTER(_ModelDesc), C.POINTER(vp)]
        L.fsmc_model_destroy.argtypes = [vp]
        L.fsmc_model_destroy.restype = None
        L.fsmc_haps_upload.argtypes = [vp, vp, u32, u32]
        L.fsmc_worklist_upload.argtypes = [vp, vp, sz, vp, sz]
        L.fsmc_decode_ibd_launch.argtypes = [vp, vp, u32]
        L.fsmc_decode_ibd_fetch.argtypes = [vp, vp, sz, C.POINTER(sz)]
        L.fsmc_sync.argtypes = [vp]
        L.fsmc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.fsmc_phase_cycles.argtypes = [vp, vp, sz]
        L.fsmc_decode_ibd.argtypes = [vp, vp, vp, sz, vp, sz, u32, vp, sz, C.POINTER(sz)]
        L.fsmc_decode_posteriors.argtypes = [vp, vp, vp, sz]
        L.fsmc_decode_per_pair.argtypes = [vp, vp, vp, vp, vp]
        L.fsmc_decode_sums.argtypes = [vp, vp, vp, vp, vp, vp]
        L.fsmc_decode_sums_batches.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp]
        L.fsmc_identify.argtypes = [vp, vp, u32, u32, vp, C.POINTER(_JobWindow), vp, u32, i32, C.c_float, C.c_float, vp,
                                    sz, C.POINTER(sz)]
        L.fsmc_identify_ex.argtypes = [vp, vp, u32, u32, vp, C.POINTER(_JobWindow), vp, u32, i32, C.c_float, C.c_float,
                                       C.POINTER(_IdentifyOpts), vp, sz, C.POINTER(sz)]
        L.fsmc_identify_fetch.argtypes = [vp, vp, sz, C.POINTER(sz)]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def whole_sequence_groups(n_pairs: int, S: int, batch: int = 64) -> np.ndarray:
    """Groups for non-hashing mode: consecutive batches, every window the whole sequence
    (reference: fromBatch = 0, toBatch = sequenceLength, HMM.cpp:87-90)."""
    n_groups = (n_pairs + batch - 1) // batch
    g = np.zeros(n_groups, GROUP_DTYPE)
    g["first_pair"] = np.arange(n_groups, dtype=np.uint64) * batch
    g["n_pairs"] = np.minimum(batch, n_pairs - g["first_pair"].astype(np.int64))
    g["from"] = 0
    g["to"] = S
    g["scan_from"] = 0
    g["scan_to"] = S
    return g


class Context:
    """One device context (``fsmc_ctx``).  ``stream`` may be a raw ``hipStream_t`` value
    (e.g. ``torch.cuda.current_stream().cuda_stream``)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._L = load()
        h = C.c_void_p()
        rc = self._L.fsmc_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise FsmcError(rc, (self._L.fsmc_last_error(None) or b"").decode())
        self._h = h
        self._models = []

    def _check(self, rc: int):
        if rc != 0:
            raise FsmcError(rc, (self._L.fsmc_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            for m in list(self._models):
                m.close()
            self._L.fsmc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        cu, slots, hbm = C.c_int32(), C.c_int32(), C.c_uint64()
        self._check(self._L.fsmc_ctx_info(self._h, C.byref(cu), C.byref(slots), C.byref(hbm)))
        chunk, chunks = C.c_int32(), C.c_int32()
        self._check(self._L.fsmc_ctx_last_plan(self._h, C.byref(chunk), C.byref(chunks), None))
        return {"n_cu": cu.value, "n_slots": slots.value, "hbm_bytes": hbm.value, "chunk_sites": chunk.value,
                "max_chunks": chunks.value}

    def set_chunk_sites(self, sites: int):
        self._check(self._L.fsmc_ctx_set_chunk_sites(self._h, sites))

    def set_resident_chunks(self, chunks: int):
        """-1 = as many of a chunked window's first chunks as memory allows keep their rows (no rebuild), 0 = none."""
        self._check(self._L.fsmc_ctx_set_resident_chunks(self._h, chunks))

    def last_resident_chunks(self) -> int:
        v = C.c_int32(0)
        self._check(self._L.fsmc_ctx_last_resident_chunks(self._h, C.byref(v)))
        return v.value

    def set_beta_stride(self, stride: int):
        """0 = automatic, 1 = every beta row through HBM, 2 = every second row (the others recomputed)."""
        self._check(self._L.fsmc_ctx_set_beta_stride(self._h, stride))

    def last_beta_stride(self) -> int:
        v = C.c_int32(0)
        self._check(self._L.fsmc_ctx_last_beta_stride(self._h, C.byref(v)))
        return v.value

    def last_kernel(self) -> int:
        """Kernel of the last launch: KT (16 ... 128) = the lane-per-pair family member, 1000 + KH (1048, 1064) = the
        four-waves-per-group kernel with KH states per wave."""
        v = C.c_int32(0)
        self._check(self._L.fsmc_ctx_last_kernel(self._h, C.byref(v)))
        return v.value

    def set_two_wave_windows(self, mode: int):
        """0 = automatic (small launches of the dump / per-pair / sums consumers give a window two waves), 1 = never."""
        self._check(self._L.fsmc_ctx_set_two_wave_windows(self._h, mode))

    def last_waves_per_window(self) -> int:
        v = C.c_int32(0)
        self._check(self._L.fsmc_ctx_last_waves_per_window(self._h, C.byref(v)))
        return v.value

    def set_pairing(self, mode: int):
        """1 (default): half-full groups with nearby windows share a wavefront; 0: groups run as uploaded."""
        self._check(self._L.fsmc_ctx_set_pairing(self._h, mode))

    def last_items(self) -> int:
        v = C.c_int32(0)
        self._check(self._L.fsmc_ctx_last_items(self._h, C.byref(v)))
        return v.value

    def last_segment_sums_in_lds(self) -> bool:
        """The last IBD launch kept the open segments' per-state sums in LDS (a launch smaller than the chip)."""
        v = C.c_int32(0)
        self._check(self._L.fsmc_ctx_last_segment_sums_in_lds(self._h, C.byref(v)))
        return bool(v.value)

    def set_workspace_limit(self, nbytes: int):
        self._check(self._L.fsmc_ctx_set_workspace_limit(self._h, nbytes))

    def expect_work(self, pair_sites: float, states: int):
        """Announce the job (pair-sites the coming launches decode): its workspace credit is there at the first launch."""
        self._check(self._L.fsmc_ctx_expect_work(self._h, float(pair_sites), int(states)))

    def create_model(self, pm) -> "Model":
        """pm: any object with the PreparedModel fields (K, S, pi, col_ratios, exp_times, D, B, U, RR,
        step_row, e1, e0m1, e2m0, state_threshold, age_threshold, probability_threshold)."""
        return Model(self, pm)

    def upload_haps(self, bits: np.ndarray, n_sites: int):
        b = np.ascontiguousarray(bits, dtype=np.uint64)
        assert b.ndim == 2 and b.shape[1] == (n_sites + 63) // 64
        self._check(self._L.fsmc_haps_upload(self._h, _p(b), b.shape[0], n_sites))

    def upload_worklist(self, pairs: np.ndarray, groups: np.ndarray):
        pr = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        gr = np.ascontiguousarray(groups, dtype=GROUP_DTYPE)
        self._check(self._L.fsmc_worklist_upload(self._h, _p(pr), pr.size, _p(gr), gr.size))
        self._n_pairs = pr.size
        self._groups = gr

    def decode_ibd_launch(self, model: "Model", flags: int = FSMC_WANT_MEAN | FSMC_WANT_MAP):
        self._check(self._L.fsmc_decode_ibd_launch(self._h, model._h, flags))

    def decode_ibd_fetch(self) -> np.ndarray:
        cap = max(1024, 4 * getattr(self, "_n_pairs", 256))
        while True:
            out = np.zeros(cap, IBD_DTYPE)
            n = C.c_size_t()
            rc = self._L.fsmc_decode_ibd_fetch(self._h, _p(out), cap, C.byref(n))
            if rc == FSMC_EOVERFLOW:
                cap = int(n.value)
                continue
            self._check(rc)
            return out[: n.value]

    def sync(self):
        self._check(self._L.fsmc_sync(self._h))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._check(self._L.fsmc_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def phase_cycles(self) -> np.ndarray:
        out = np.zeros(128, np.uint64)  # kPhaseSlots (fsmc_kernels.h)
        if self._L.fsmc_phase_cycles(self._h, _p(out), 128) != 0:  # (a library of an earlier round: 32 slots)
            self._check(self._L.fsmc_phase_cycles(self._h, _p(out), 32))
        return out

    def decode_ibd(self, model: "Model", pairs, groups, flags: int = FSMC_WANT_MEAN | FSMC_WANT_MAP) -> np.ndarray:
        self.upload_worklist(pairs, groups)
        self.decode_ibd_launch(model, flags)
        return self.decode_ibd_fetch()

    def identify(self, words, global_ids, gen_pos, *, window_size=0, w_i=1, w_j=1, last_job=True, j_above_diag=False,
                 gap=1, skip=0.0, min_m=1.0, word_size=64, haploid=True, max_seeds=0, read_ahead=10) -> np.ndarray:
        """The identification step (fsmc_identify_ex): candidates (hap_a, hap_b, from, to, flush_word) in emission
        order.  ``words``: uint64 [n_haps][n_words]; ``global_ids``: haplotype numbers in the whole file; ``gen_pos``:
        Morgans per site.  The defaults of the job window are those of a single job (every pair belongs to it); the
        defaults of the last four are the reference's (DecodingParams.hpp)."""
        w = np.ascontiguousarray(words, np.uint64)
        if w.ndim != 2:
            raise ValueError("words must be [n_haps][n_words]")
        ids = np.ascontiguousarray(global_ids, np.uint32)
        gen = np.ascontiguousarray(gen_pos, np.float32)
        if ids.shape != (w.shape[0],):
            raise ValueError("global_ids must have one entry per haplotype")
        jw = _JobWindow(window_size, w_i, w_j, int(bool(last_job)), int(bool(j_above_diag)))
        opts = _IdentifyOpts(word_size, int(bool(haploid)), max_seeds, read_ahead)
        cap = max(1024, 4 * w.shape[0])
        while True:
            out = np.zeros(cap, CANDIDATE_DTYPE)
            n = C.c_size_t(0)
            rc = self._L.fsmc_identify_ex(self._h, _p(w), w.shape[0], w.shape[1], _p(ids), C.byref(jw), _p(gen),
                                          gen.size, gap, skip, min_m, C.byref(opts), _p(out), cap, C.byref(n))
            if rc == -6:  # FSMC_EOVERFLOW: n holds the count, the finished list waits on the device
                cap = int(n.value)
                out = np.zeros(cap, CANDIDATE_DTYPE)
                if self._L.fsmc_identify_fetch(self._h, _p(out), cap, C.byref(n)) == 0:
                    return out[:n.value]
                continue
            self._check(rc)
            return out[:n.value]

    def decode_per_pair(self, model: "Model", exp_coal_times, want_mean=True, want_map=True):
        """writePerPairOutput for the resident work list: (mean[n_pairs][S] f32, map[n_pairs][S] i32)."""
        et = np.ascontiguousarray(exp_coal_times, np.float32)
        mean = np.zeros((self._n_pairs, model.S), np.float32) if want_mean else None
        mp = np.zeros((self._n_pairs, model.S), np.int32) if want_map else None
        self._check(self._L.fsmc_decode_per_pair(self._h, model._h, _p(et), _p(mean), _p(mp)))
        return mean, mp

    def decode_sums(self, model: "Model", major_minor: bool = False, sums: bool = True, into=None, batch_first_group=None):
        """augmentSumOverPairs for the resident work list: arrays [S][K] (sum, and 00/01/11 when asked).  ``into`` =
        (sum, [s00, s01, s11]) of an earlier call continues that accumulation (sumOverPairs += ..., HMM.cpp:1073).
        ``batch_first_group`` (n_batches + 1 entries): the reference batches, as runs of consecutive groups, when a
        batch holds more than 64 pairs (fsmc_decode_sums_batches)."""
        shape = (model.S, model.K)
        if into is not None:
            s, mm = into
            mm = list(mm) if major_minor else [None, None, None]
        else:
            s = np.zeros(shape, np.float32) if sums else None
            mm = [np.zeros(shape, np.float32) for _ in range(3)] if major_minor else [None, None, None]
        if batch_first_group is not None:
            bf = np.ascontiguousarray(batch_first_group, np.uint32)
            self._check(self._L.fsmc_decode_sums_batches(self._h, model._h, _p(bf), bf.size - 1, _p(s), _p(mm[0]),
                                                         _p(mm[1]), _p(mm[2])))
        else:
            self._check(self._L.fsmc_decode_sums(self._h, model._h, _p(s), _p(mm[0]), _p(mm[1]), _p(mm[2])))
        return s, mm

    def decode_posteriors(self, model: "Model") -> list[np.ndarray]:
        """Posterior per group in the reference's batch layout: list of [to-from][K][64] arrays."""
        gr = self._groups
        sizes = [64 * model.K * int(g["to"] - g["from"]) for g in gr]
        out = np.zeros(sum(sizes), np.float32)
        self._check(self._L.fsmc_decode_posteriors(self._h, model._h, _p(out), out.size))
        res, o = [], 0
        for g, n in zip(gr, sizes):
            res.append(out[o:o + n].reshape(int(g["to"] - g["from"]), model.K, 64))
            o += n
        return res


class Model:
    def __init__(self, ctx: Context, pm):
        self._ctx = ctx
        self.K, self.S = int(pm.K), int(pm.S)
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
        keep = dict(pi=f32(pm.pi), cr=f32(pm.col_ratios), et=f32(pm.exp_times), D=f32(pm.D), B=f32(pm.B),
                    U=f32(pm.U), RR=f32(pm.RR), sr=np.ascontiguousarray(pm.step_row, dtype=np.int32),
                    e1=f32(pm.e1), e0m1=f32(pm.e0m1), e2m0=f32(pm.e2m0))
        # the C side reads these with the shapes of fsmc_model_desc: a mis-shaped input must not be read out of bounds
        rows = keep["D"].shape[0] if keep["D"].ndim == 2 else -1
        for name, want in (("pi", (self.K,)), ("cr", (self.K,)), ("et", (self.K,)), ("D", (rows, self.K)),
                           ("B", (rows, self.K)), ("U", (rows, self.K)), ("RR", (rows, self.K)), ("sr", (self.S,)),
                           ("e1", (self.S, self.K)), ("e0m1", (self.S, self.K)), ("e2m0", (self.S, self.K))):
            if keep[name].shape != want:
                raise ValueError(f"model field {name}: shape {keep[name].shape}, expected {want}")
        d = _ModelDesc()
        d.K, d.S = self.K, self.S
        d.pi, d.col_ratios, d.exp_times = _p(keep["pi"]), _p(keep["cr"]), _p(keep["et"])
        d.n_rows = keep["D"].shape[0]
        d.D, d.B, d.U, d.RR = _p(keep["D"]), _p(keep["B"]), _p(keep["U"]), _p(keep["RR"])
        d.step_row = _p(keep["sr"])
        d.e1, d.e0m1, d.e2m0 = _p(keep["e1"]), _p(keep["e0m1"]), _p(keep["e2m0"])
        d.state_threshold = int(pm.state_threshold)
        d.age_threshold = int(pm.age_threshold)
        d.probability_threshold = float(pm.probability_threshold)
        if getattr(pm, "sequence", False):  # decodingSequence: two steps per site (fsmc_model_desc)
            i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
            keep.update(gf=i32(pm.gap_row_f), sf=i32(pm.site_row_f), gb=i32(pm.gap_row_b), sb=i32(pm.site_row_b),
                        hom=f32(pm.hom))
            for name, want in (("gf", (self.S,)), ("sf", (self.S,)), ("gb", (self.S,)), ("sb", (self.S,)),
                               ("hom", (self.S, self.K))):
                if keep[name].shape != want:
                    raise ValueError(f"model field {name}: shape {keep[name].shape}, expected {want}")
            d.sequence = 1
            d.gap_row_f, d.site_row_f = _p(keep["gf"]), _p(keep["sf"])
            d.gap_row_b, d.site_row_b = _p(keep["gb"]), _p(keep["sb"])
            d.hom = _p(keep["hom"])
        h = C.c_void_p()
        ctx._check(ctx._L.fsmc_model_create(ctx._h, C.byref(d), C.byref(h)))
        self._h = h
        ctx._models.append(self)

    def close(self):
        if getattr(self, "_h", None):
            self._ctx._L.fsmc_model_destroy(self._h)
            self._h = None
            if self in self._ctx._models:
                self._ctx._models.remove(self)
