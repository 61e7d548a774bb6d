"""ctypes front-end of the CPU oracle (oracle/liboracle.so) plus the host-side orchestration
of the reference restated in numpy.

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and by
``bench.py``'s ``cpu_baseline`` leg -- never by the product package ``fastsmc_amd``.

Citations are to /root/reference/ASMC_SRC/SRC.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_LIB_PATHS = {"ref": _LIB_PATH, "avx2": os.path.join(_HERE, "liboracle_avx2.so")}
_variant = "ref"
_libs: dict = {}


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc; see oracle/Makefile): liboracle.so (-O2, the checker) and
    liboracle_avx2.so (-O3 -mavx2, same sources; only bench.py's cpu_baseline times it)."""
    if force or not all(os.path.exists(q) for q in _LIB_PATHS.values()):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def select_build(name: str) -> None:
    """Which build lib() hands out: "ref" (default, the parity checker) or "avx2" (timing only)."""
    global _variant, _lib
    assert name in _LIB_PATHS
    _variant = name
    _lib = _libs.get(name)


class _FoModel(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("S", C.c_int32),
        ("pi", C.c_void_p), ("colRatios", C.c_void_p), ("expTimes", C.c_void_p),
        ("nRows", C.c_int32),
        ("Dt", C.c_void_p), ("Bt", C.c_void_p), ("Ut", C.c_void_p), ("RRt", C.c_void_p),
        ("stepRow", C.c_void_p),
        ("e1", C.c_void_p), ("e0m1", C.c_void_p), ("e2m0", C.c_void_p),
        ("sequence", C.c_int32),
        ("gapRowF", C.c_void_p), ("siteRowF", C.c_void_p), ("gapRowB", C.c_void_p), ("siteRowB", C.c_void_p),
        ("hom", C.c_void_p),
    ]


class IbdRecord(C.Structure):
    _fields_ = [("pair", C.c_uint32), ("start", C.c_int32), ("end", C.c_int32), ("prob", C.c_float),
                ("postMean", C.c_float), ("map", C.c_float)]


IBD_DTYPE = np.dtype([("pair", "<u4"), ("start", "<i4"), ("end", "<i4"), ("prob", "<f4"), ("postMean", "<f4"),
                      ("map", "<f4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATHS[_variant])
        L.fo_round_morgans.restype = C.c_float
        L.fo_round_morgans.argtypes = [C.c_float, C.c_int, C.c_float]
        L.fo_round_physical.restype = C.c_int
        L.fo_round_physical.argtypes = [C.c_int, C.c_int]
        L.fo_get_from_position.restype = C.c_uint
        L.fo_get_from_position.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_float]
        L.fo_get_to_position.restype = C.c_uint
        L.fo_get_to_position.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_float]
        L.fo_subset_xor.restype = C.c_ulong
        L.fo_subset_xor.argtypes = [C.c_void_p, C.c_void_p, C.c_ulong, C.c_ulong, C.c_ulong, C.c_void_p]
        L.fo_subset_and.restype = C.c_ulong
        L.fo_subset_and.argtypes = [C.c_void_p, C.c_void_p, C.c_ulong, C.c_ulong, C.c_ulong, C.c_void_p]
        L.fo_calculate_scaling_batch.restype = None
        L.fo_calculate_scaling_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.fo_apply_scaling_batch.restype = None
        L.fo_apply_scaling_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.fo_decode_scalar.restype = None
        L.fo_decode_scalar.argtypes = [C.POINTER(_FoModel), C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
        L.fo_decode_batch.restype = None
        L.fo_decode_batch.argtypes = [C.POINTER(_FoModel), C.c_void_p, C.c_void_p, C.c_int, C.c_uint, C.c_uint,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.fo_augment_sum_over_pairs.restype = None
        L.fo_augment_sum_over_pairs.argtypes = [C.POINTER(_FoModel), C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p]
        L.fo_per_pair_output.restype = None
        L.fo_per_pair_output.argtypes = [C.POINTER(_FoModel), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.fo_posterior_mean.restype = C.c_float
        L.fo_posterior_mean.argtypes = [C.POINTER(_FoModel), C.c_void_p, C.c_int]
        L.fo_map.restype = C.c_float
        L.fo_map.argtypes = [C.POINTER(_FoModel), C.c_void_p, C.c_int]
        L.fo_ibd_scan_pair.restype = C.c_int
        L.fo_ibd_scan_pair.argtypes = [C.POINTER(_FoModel), C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_uint,
                                       C.c_uint, C.c_uint, C.c_float, C.c_int, C.c_int, C.c_uint32, C.c_void_p,
                                       C.c_int]
        L.fo_undistinguished_counts.restype = C.c_int
        L.fo_undistinguished_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_void_p]
        _lib = L
        _libs[_variant] = L
    return _lib


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ helpers

def round_morgans(value: float, precision: int = 2, min_: float = 1e-10) -> np.float32:
    return np.float32(lib().fo_round_morgans(np.float32(value), precision, np.float32(min_)))


def round_physical(value: int, precision: int = 2) -> int:
    return int(lib().fo_round_physical(int(value), precision))


def get_from_position(gen: np.ndarray, frm: int, cm_dist: float = 0.5) -> int:
    g = np.ascontiguousarray(gen, np.float32)
    return int(lib().fo_get_from_position(_p(g), g.size, frm, np.float32(cm_dist)))


def get_to_position(gen: np.ndarray, to: int, cm_dist: float = 0.5) -> int:
    g = np.ascontiguousarray(gen, np.float32)
    return int(lib().fo_get_to_position(_p(g), g.size, to, np.float32(cm_dist)))


def subset_xor(v1, v2, frm=0, to=None):
    a = np.ascontiguousarray(v1, np.uint8)
    b = np.ascontiguousarray(v2, np.uint8)
    to = (1 << 62) if to is None else to
    out = np.zeros(max(0, min(a.size, to) - frm), np.uint8)
    n = lib().fo_subset_xor(_p(a), _p(b), a.size, frm, to, _p(out))
    return out[:n]


def subset_and(v1, v2, frm=0, to=None):
    a = np.ascontiguousarray(v1, np.uint8)
    b = np.ascontiguousarray(v2, np.uint8)
    to = (1 << 62) if to is None else to
    out = np.zeros(max(0, min(a.size, to) - frm), np.uint8)
    n = lib().fo_subset_and(_p(a), _p(b), a.size, frm, to, _p(out))
    return out[:n]


def calculate_scaling_batch(vec: np.ndarray, batch: int, states: int):
    v = np.ascontiguousarray(vec, np.float32)
    scal = np.zeros(batch, np.float32)
    sums = np.zeros(batch, np.float32)
    lib().fo_calculate_scaling_batch(_p(v), _p(scal), _p(sums), batch, states)
    return scal, sums


def apply_scaling_batch(vec: np.ndarray, scal: np.ndarray, batch: int, states: int):
    v = np.array(vec, np.float32, copy=True)
    s = np.ascontiguousarray(scal, np.float32)
    lib().fo_apply_scaling_batch(_p(v), _p(s), batch, states)
    return v


def undistinguished_counts(derived, total, csfs_samples, fold=True, uses_csfs=True, known_seed=True):
    """Data::calculateUndistinguishedCounts after std::srand(1234) (Data.cpp:55-60, 567-599)."""
    d = np.ascontiguousarray(derived, np.int32)
    t = np.ascontiguousarray(total, np.int32)
    out = np.zeros((d.size, 3), np.int32)
    rc = lib().fo_undistinguished_counts(_p(d), _p(t), d.size, csfs_samples, int(fold), int(uses_csfs),
                                         int(known_seed), _p(out))
    if rc != 0:
        raise RuntimeError("undistinguished counts: a site violates the reference's checks")
    return out


# ------------------------------------------------------ prepared model (HMM ctor)

@dataclass
class PreparedModel:
    """Constant inputs of the decode path, as the HMM constructor leaves them
    (HMM.cpp:65-127, prepareEmissions 159-256)."""

    K: int
    S: int
    pi: np.ndarray
    col_ratios: np.ndarray
    exp_times: np.ndarray
    D: np.ndarray
    B: np.ndarray
    U: np.ndarray
    RR: np.ndarray
    step_row: np.ndarray  # [S] int32
    e1: np.ndarray  # [S][K]
    e0m1: np.ndarray
    e2m0: np.ndarray
    gen: np.ndarray  # geneticPositions [S] f32 (Morgans)
    phys: np.ndarray  # physicalPositions [S] i32
    state_threshold: int
    age_threshold: int
    probability_threshold: np.float32
    # sequence mode (decodingSequence): two steps per site, see hmm_oracle.h
    sequence: bool = False
    gap_row_f: np.ndarray | None = None  # [S] int32
    site_row_f: np.ndarray | None = None
    gap_row_b: np.ndarray | None = None
    site_row_b: np.ndarray | None = None
    hom: np.ndarray | None = None  # [S][K] f32

    def c_struct(self) -> _FoModel:
        m = _FoModel()
        m.K, m.S = self.K, self.S
        m.pi, m.colRatios, m.expTimes = _p(self.pi), _p(self.col_ratios), _p(self.exp_times)
        m.nRows = self.D.shape[0]
        m.Dt, m.Bt, m.Ut, m.RRt = _p(self.D), _p(self.B), _p(self.U), _p(self.RR)
        m.stepRow = _p(self.step_row)
        m.e1, m.e0m1, m.e2m0 = _p(self.e1), _p(self.e0m1), _p(self.e2m0)
        m.sequence = int(self.sequence)
        if self.sequence:
            m.gapRowF, m.siteRowF = _p(self.gap_row_f), _p(self.site_row_f)
            m.gapRowB, m.siteRowB = _p(self.gap_row_b), _p(self.site_row_b)
            m.hom = _p(self.hom)
        return m


def step_rows(tables_keys: np.ndarray, gen: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Per-site transition-table row: key = roundMorgans(gen[p]-gen[p-1], 2, 1e-10f) in fp32
    (HMM.cpp:755, 909; HMM.hpp:129-130), looked up by exact float match (HMM.cpp:795-797).
    Returns (step_row[S] into the *full* key table, keys[S])."""
    gen = np.ascontiguousarray(gen, np.float32)
    key_to_row = {np.float32(k).tobytes(): i for i, k in enumerate(np.asarray(tables_keys, np.float32))}
    S = gen.size
    rows = np.zeros(S, np.int32)
    keys = np.zeros(S, np.float32)
    for p in range(1, S):
        k = round_morgans(np.float32(gen[p] - gen[p - 1]))
        keys[p] = k
        r = key_to_row.get(np.float32(k).tobytes())
        if r is None:  # the reference throws std::out_of_range from unordered_map::at
            raise KeyError(f"no transition vectors for genetic distance key {k!r} (site {p})")
        rows[p] = r
    return rows, keys


def prepare_model(tables, gen, phys, derived_counts, total_samples: int, *, time: int = 50,
                  no_conditional_age_estimates: bool = True, fold: bool = True, decoding_sequence: bool = False,
                  skip_csfs_distance: float = 0.0, known_seed: bool = True, compact_rows: bool = True,
                  rec_rate: np.ndarray | None = None) -> PreparedModel:
    """HMM::HMM + prepareEmissions (folded or not, array or sequence mode), HMM.cpp:65-127, 159-256.

    ``tables`` is a fastsmc_amd.synth.ModelTables-like object (fields of DecodingQuantities).
    ``rec_rate``: Data::recRateAtMarker; default = the FastSMC-mode reader's values (rec_rate_at_marker)."""
    K = int(tables.K)
    gen = np.ascontiguousarray(gen, np.float32)
    S = gen.size
    und = undistinguished_counts(derived_counts, np.full(S, total_samples, np.int32), tables.csfs_samples, fold=fold,
                                 uses_csfs=True, known_seed=known_seed)
    use_csfs = np.zeros(S, bool)
    if skip_csfs_distance < np.inf:  # HMM.cpp:163-173
        use_csfs[0] = True
        last = np.float32(0.0)
        skip = np.float32(skip_csfs_distance)
        for pos in range(1, S):
            if np.float32(gen[pos] - last) >= skip:
                use_csfs[pos] = True
                last = gen[pos]
    e1 = np.zeros((S, K), np.float32)
    e0m1 = np.zeros((S, K), np.float32)
    e2m0 = np.zeros((S, K), np.float32)
    # sequence data: CSFS / folded CSFS / classic emission; array data: the ascertained ones (HMM.cpp:183-252)
    FAC = tables.folded_csfs if decoding_sequence else tables.folded_ascertained_csfs
    ASC = tables.csfs if decoding_sequence else tables.ascertained_csfs
    comp = tables.classic_emission if decoding_sequence else tables.compressed_emission
    for pos in range(S):
        if use_csfs[pos]:
            u0, u1, u2 = (int(x) for x in und[pos])
            if fold:  # HMM.cpp:179-207 (array: foldedAscertainedCSFSmap)
                e1[pos] = FAC[u1][1] if u1 >= 0 else 0.0
                e0m1[pos] = FAC[u0][0] - e1[pos]
                e2m0[pos] = (FAC[u2][0] - FAC[u0][0]) if u2 >= 0 else (np.float32(0) - FAC[u0][0])
            else:  # HMM.cpp:208-240 (array: ascertainedCSFSmap)
                e1[pos] = ASC[u1][1] if u1 >= 0 else 0.0
                e0 = ASC[u0][0] if u0 >= 0 else np.zeros(K, np.float32)
                e0m1[pos] = e0 - e1[pos]
                if u2 >= 0:
                    dist, undist = (0, 0) if u2 == tables.csfs_samples - 2 else (2, u2)
                    e2m0[pos] = ASC[undist][dist] - e0
                else:
                    e2m0[pos] = np.float32(0) - e0
        else:  # HMM.cpp:242-253
            e1[pos] = comp[1]
            e0m1[pos] = comp[0] - comp[1]
            e2m0[pos] = 0.0
    full_rows, step_keys = step_rows(tables.keys, gen)
    seq_rows = []
    hom = None
    if decoding_sequence:
        # HMM.cpp:752-770 (forward) and 905-925 (backward)
        phys_i = np.ascontiguousarray(phys, np.int64)
        rate = rec_rate_at_marker(gen, phys_i) if rec_rate is None else np.ascontiguousarray(rec_rate, np.float32)
        key_to_row = {np.float32(k).tobytes(): i for i, k in enumerate(np.asarray(tables.keys, np.float32))}

        def row_of(key, what, site):
            r = key_to_row.get(np.float32(key).tobytes())
            if r is None:
                raise KeyError(f"no transition vectors for {what} key {key!r} (site {site})")
            return r

        gf, sf, gb, sb = (np.zeros(S, np.int32) for _ in range(4))
        hom = np.zeros((S, K), np.float32)
        hom_row = {int(k): i for i, k in enumerate(tables.homozygous_keys)}
        for q in range(1, S):
            rec_dist = step_keys[q]
            rate_q, rate_p = round_morgans(rate[q]), round_morgans(rate[q - 1])
            gf[q] = row_of(round_morgans(np.float32(rec_dist - rate_q)), "forward gap", q)
            sf[q] = row_of(rate_q, "forward site", q)
            gb[q] = row_of(round_morgans(np.float32(rec_dist - rate_p)), "backward gap", q)
            sb[q] = row_of(rate_p, "backward site", q)
            d = round_physical(int(phys_i[q] - phys_i[q - 1] - 1))
            if d not in hom_row:  # unordered_map::at
                raise KeyError(f"no homozygous emission for physical distance {d} (site {q})")
            hom[q] = tables.homozygous[hom_row[d]]
        seq_rows = [gf, sf, gb, sb]
    if compact_rows:
        used, inv = np.unique(np.concatenate([full_rows] + seq_rows), return_inverse=True)
        inv = inv.astype(np.int32).reshape(1 + len(seq_rows), S)
        rows, seq_rows = inv[0].copy(), [r.copy() for r in inv[1:]]
        D, B, U, RR = (np.ascontiguousarray(x[used]) for x in (tables.D, tables.B, tables.U, tables.RR))
    else:
        rows = full_rows
        D, B, U, RR = tables.D, tables.B, tables.U, tables.RR
    # HMM::getStateThreshold (HMM.cpp:504-513) and the probability threshold (96-99), fp32 accumulation
    disc = tables.discretization
    st = 0
    while st < K and disc[st] < np.float32(time):
        st += 1
    pthr = np.float32(0.0)
    for i in range(st):
        pthr = np.float32(pthr + tables.initial_state_prob[i])
    return PreparedModel(
        K=K, S=S, pi=np.ascontiguousarray(tables.initial_state_prob, np.float32),
        col_ratios=np.ascontiguousarray(tables.column_ratios, np.float32),
        exp_times=np.ascontiguousarray(tables.expected_times, np.float32),
        D=np.ascontiguousarray(D, np.float32), B=np.ascontiguousarray(B, np.float32),
        U=np.ascontiguousarray(U, np.float32), RR=np.ascontiguousarray(RR, np.float32),
        step_row=np.ascontiguousarray(rows, np.int32), e1=e1, e0m1=e0m1, e2m0=e2m0, gen=gen,
        phys=np.ascontiguousarray(phys, np.int32), state_threshold=st,
        age_threshold=K if no_conditional_age_estimates else st, probability_threshold=pthr,
        sequence=bool(decoding_sequence),
        **(dict(gap_row_f=seq_rows[0], site_row_f=seq_rows[1], gap_row_b=seq_rows[2], site_row_b=seq_rows[3],
                hom=hom) if decoding_sequence else {}))


def rec_rate_at_marker(gen: np.ndarray, phys: np.ndarray) -> np.ndarray:
    """Data::recRateAtMarker as the FastSMC-mode reader fills it (Data::addMarker, Data.cpp:549-565): the fp32
    difference of genetic positions, promoted to double, divided by the integer distance, rounded to fp32; marker 0
    takes the rate of marker 1."""
    gen = np.ascontiguousarray(gen, np.float32)
    phys = np.ascontiguousarray(phys, np.int64)
    out = np.zeros(gen.size, np.float32)
    for p in range(1, gen.size):
        out[p] = np.float32(float(np.float32(gen[p] - gen[p - 1])) / float(int(phys[p] - phys[p - 1])))
    if gen.size > 1:
        out[0] = out[1]
    return out


# ------------------------------------------------------------ the hot path

def decode_batch(model: PreparedModel, obs_bits: np.ndarray, hom_bits: np.ndarray, frm: int, to: int,
                 want_alpha_fwd: bool = False, buffers=None):
    """HMM::decodeBatch. obs_bits/hom_bits: [B][to-frm] uint8. Returns (posterior, beta[, alphaFwd]) with the
    reference layout [S][K][B]; only rows [frm, to) are meaningful.  buffers: optional (alpha, beta) arrays of
    that shape to decode into (the reference reuses its m_alphaBuffer / m_betaBuffer from batch to batch too)."""
    ob = np.ascontiguousarray(obs_bits, np.uint8)
    hb = np.ascontiguousarray(hom_bits, np.uint8)
    B = ob.shape[0]
    assert ob.shape == hb.shape == (B, to - frm)
    if buffers is not None:
        alpha, beta = buffers
        assert alpha.shape == beta.shape == (model.S, model.K, B) and alpha.dtype == beta.dtype == np.float32
    else:
        alpha = np.zeros((model.S, model.K, B), np.float32)
        beta = np.zeros((model.S, model.K, B), np.float32)
    afwd = np.zeros((model.S, model.K, B), np.float32) if want_alpha_fwd else None
    m = model.c_struct()
    lib().fo_decode_batch(C.byref(m), _p(ob), _p(hb), B, frm, to, _p(alpha), _p(beta), _p(afwd))
    return (alpha, beta, afwd) if want_alpha_fwd else (alpha, beta)


def decode_scalar(model: PreparedModel, obs_bits: np.ndarray, hom_bits: np.ndarray, frm: int = 0, to: int | None = None):
    """The reference's scalar path for ONE pair (HMM::decode, HMM.cpp:1469-1495: forward 1533-1609, backward
    1636-1690, product and column normalisation): the reference's own second implementation of the chain.
    obs_bits / hom_bits: [S] uint8, absolute sites.  Returns the posterior [K][S] (columns outside [frm, to) zero)."""
    ob = np.ascontiguousarray(obs_bits, np.uint8)
    hb = np.ascontiguousarray(hom_bits, np.uint8)
    to = model.S if to is None else to
    assert ob.shape == hb.shape == (model.S,) and 0 <= frm < to <= model.S
    post = np.zeros((model.K, model.S), np.float32)
    m = model.c_struct()
    lib().fo_decode_scalar(C.byref(m), _p(ob), _p(hb), frm, to, _p(post))
    return post


def ibd_scan_pair(model: PreparedModel, post: np.ndarray, v: int, frm: int, to: int, *, want_mean=True,
                  want_map=True, pair_ordinal=0, cap=4096) -> np.ndarray:
    out = np.zeros(cap, IBD_DTYPE)
    m = model.c_struct()
    n = lib().fo_ibd_scan_pair(C.byref(m), _p(post), post.shape[2], v, frm, to, model.state_threshold,
                               model.age_threshold, model.probability_threshold, int(want_mean), int(want_map),
                               pair_ordinal, _p(out), cap)
    if n > cap:
        return ibd_scan_pair(model, post, v, frm, to, want_mean=want_mean, want_map=want_map,
                             pair_ordinal=pair_ordinal, cap=n)
    return out[:n]


def augment_sum_over_pairs(model: PreparedModel, post, actual_b, obs_bits, hom_bits, sums, sums00=None, sums01=None,
                           sums11=None):
    m = model.c_struct()
    ob = np.ascontiguousarray(obs_bits, np.uint8)
    hb = np.ascontiguousarray(hom_bits, np.uint8)
    lib().fo_augment_sum_over_pairs(C.byref(m), _p(post), actual_b, post.shape[2], _p(ob), _p(hb),
                                    int(sums is not None), int(sums00 is not None), _p(sums), _p(sums00), _p(sums01),
                                    _p(sums11))


def per_pair_output(model: PreparedModel, post, actual_b, *, want_post=False, sum_of_post=None):
    m = model.c_struct()
    mean = np.zeros((actual_b, model.S), np.float32)
    MAP = np.zeros((actual_b, model.S), np.int32)
    pp = np.zeros((actual_b, model.K, model.S), np.float32) if want_post else None
    lib().fo_per_pair_output(C.byref(m), _p(post), actual_b, post.shape[2], _p(model.exp_times), _p(mean), _p(MAP),
                             _p(pp), _p(sum_of_post))
    return mean, MAP, pp


# -------------------------------------------------- orchestration (HMM.cpp:283-636)

def enumerate_all_pairs(n_ind: int, jobs: int = 1, job_ind: int = 1, within_only: bool = False):
    """Pair order of HMM::decodeAll (HMM.cpp:310-364): for i, for j<i, iHap, jHap -> makePairObs(jHap, j, iHap, i);
    then (1, i, 2, i).  Returns a list of (hapRowA, hapRowB) global haplotype rows with A the record's first id."""
    tot = n_ind if within_only else 2 * n_ind * n_ind - n_ind
    start = tot * (job_ind - 1) // jobs
    end = tot * job_ind // jobs
    out = []
    pairs = 0
    for i in range(n_ind):
        if not within_only:
            for j in range(i):
                for i_hap in (1, 2):
                    for j_hap in (1, 2):
                        if start <= pairs < end:
                            out.append((2 * j + j_hap - 1, 2 * i + i_hap - 1))
                        pairs += 1
        if start <= pairs < end:
            out.append((2 * i, 2 * i + 1))
        pairs += 1
    return out


def decode_pairs_ibd(model: PreparedModel, hap_bytes: np.ndarray, pairs, *, batch_size=32, want_mean=True,
                     want_map=True, sums=None, threads: int = 1):
    """Non-hashing FastSMC-mode decode of a pair list: addToBatch/runLastBatch with whole-sequence windows
    (HMM.cpp:555-636), then writePerPairOutputFastSMC per batch.  hap_bytes: [n_hap][S] uint8 folded alleles.
    Returns IBD records in the reference's output order (batch, pair in batch, site).  threads > 1 decodes
    batches concurrently (the C calls release the GIL; batches are independent) -- the reference itself is
    single-threaded, this is only how the CPU baseline uses all host cores."""
    S = model.S
    import threading

    tls = threading.local()

    def one_batch(b0: int):
        chunk = pairs[b0:b0 + batch_size]
        actual = len(chunk)
        padded = list(chunk)
        while len(padded) % 4:  # VECX == 4 in the NO_SSE build (AvxDefinitions.hpp:22-25; HMM.cpp:617-619)
            padded.append(padded[-1])
        ob = np.stack([hap_bytes[a] ^ hap_bytes[b] for a, b in padded])
        hb = np.stack([hap_bytes[a] & hap_bytes[b] for a, b in padded])
        bufs = getattr(tls, "bufs", None)  # one alpha/beta buffer pair per thread, like the reference's members
        if bufs is None or bufs[0].shape[2] != len(padded):
            bufs = (np.zeros((S, model.K, len(padded)), np.float32), np.zeros((S, model.K, len(padded)), np.float32))
            tls.bufs = bufs
        post, _ = decode_batch(model, ob, hb, 0, S, buffers=bufs)
        if sums is not None:
            augment_sum_over_pairs(model, post, actual, ob, hb, sums)
        return [ibd_scan_pair(model, post, v, 0, S, want_mean=want_mean, want_map=want_map, pair_ordinal=b0 + v)
                for v in range(actual)]

    starts = range(0, len(pairs), batch_size)
    if threads > 1 and sums is None:
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=threads) as ex:
            per_batch = list(ex.map(one_batch, starts))
    else:
        per_batch = [one_batch(b0) for b0 in starts]
    recs = [r for batch in per_batch for r in batch]
    return np.concatenate(recs) if recs else np.zeros(0, IBD_DTYPE)


# ----------------------------------------------------- record formatting (HMM.cpp:1110-1144)

def _g7(x) -> str:
    """C++ ostream << with setprecision(7) (std::numeric_limits<float>::digits10 + 1), default float field."""
    return format(float(x), ".7g")


def eigen_format_rows(mat) -> str:
    """``matrix.format(Eigen::IOFormat(FullPrecision, DontAlignCols, " ", "\\n"))`` (HMM.hpp:154; used by
    HMM::writePerPairOutput, HMM.cpp:1412-1420): coefficients of a row separated by a blank, rows by a newline --
    BETWEEN rows: nothing follows the last one, so two matrices streamed one after the other share a line.
    FullPrecision prints a float with NumTraits<float>::digits10() = 6 significant digits in the stream's general
    notation (Eigen 3.4, what the reference's unpinned vcpkg dependency resolves to), i.e. "%.6g"; integers as they
    are."""
    a = np.asarray(mat)
    if np.issubdtype(a.dtype, np.integer):
        return "\n".join(" ".join(str(int(x)) for x in row) for row in a)
    return "\n".join(" ".join("%.6g" % float(x) for x in row) for row in a)


def format_ibd_text(recs: np.ndarray, pairs, fam_ids, iids, chr_number: int, phys, gen, *, want_length=True,
                    want_mean=True, want_map=True) -> str:
    """Text lines HMM::writePairIBD writes for these records; ``pairs[i] = (hapRowA, hapRowB)`` and a haplotype
    row r is individual r // 2, haplotype r % 2 + 1."""
    lines = []
    gen = np.asarray(gen, np.float32)
    for r in recs:
        ha, hb = pairs[int(r["pair"])]
        ia, ib = ha // 2, hb // 2
        f = [fam_ids[ia], iids[ia], str(ha % 2 + 1), fam_ids[ib], iids[ib], str(hb % 2 + 1), str(chr_number),
             str(int(phys[r["start"]])), str(int(phys[r["end"]]))]
        if want_length:
            f.append(_g7(np.float32(100.0) * np.float32(gen[r["end"]] - gen[r["start"]])))
        f.append(_g7(float(np.float32(r["prob"])) / float(int(r["end"]) - int(r["start"]) + 1)))
        if want_mean:
            f.append(_g7(r["postMean"]))
        if want_map:
            f.append(_g7(r["map"]))
        lines.append("\t".join(f) + "\n")
    return "".join(lines)


def job_individuals(sample_size: int, jobs: int, job_ind: int):
    """Individuals a job loads (Data.cpp:62-80, 251-262): windows w_i, w_j of the square job grid."""
    import math

    window = math.ceil(math.sqrt((2.0 * sample_size ** 2 - sample_size) * 2.0 / jobs))
    if window % 2:
        window += 1
    w_i, cpt_job, cpt_tot = 1, 1, 1
    while cpt_tot < job_ind:
        w_i += 1
        cpt_job += 2
        cpt_tot += cpt_job
    w_j = math.ceil(np.float32(cpt_job - (cpt_tot - job_ind)) / 2)
    keep = []
    for d in range(sample_size):
        if ((w_i - 1) * window // 2 <= d < w_i * window // 2) or ((w_j - 1) * window // 2 <= d < w_j * window // 2) \
                or (jobs == job_ind and d >= (w_j - 1) * window // 2):
            keep.append(d)
    return keep
