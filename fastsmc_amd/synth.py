"""Synthetic inputs for the decode path: decoding-quantities tables and haplotypes.

Every ``*.decodingQuantities.gz`` of the reference checkout is a missing blob
(/root/reference/.MISSING_LARGE_BLOBS), so tests and ``bench.py`` run on tables generated
here.  The tables have exactly the structure the decoder consumes (SURVEY.md App. A/B;
reference ``DecodingQuantities.hpp:51-68``): a semiseparable transition described by the
per-distance vectors ``D, B, U, rowRatios`` plus the distance-independent ``columnRatios``,
for a constant-size population (closed forms below), keyed by the reference's 3-significant-
digit genetic-distance grid (``DecodingQuantities.java:81-87,159-165``).

Nothing here is on the timed path; it is input preparation.
"""
from __future__ import annotations

import gzip
import math
from dataclasses import dataclass, field

import numpy as np

# FILES/DISC/30-100-2000.disc of the reference (69 interval starts; the last interval is open).
DISC_30_100_2000 = [
    0, 30, 60, 90, 120, 150, 180, 210, 240, 270, 300, 330, 360, 390, 420, 520, 620, 720, 820, 920, 1020, 1120,
    1220, 1320, 1420, 1520, 1620, 1720, 1820, 1920, 2053.0, 2183.0, 2331.3, 2497.0, 2653.7, 2823.9, 3010.1, 3250.9,
    3554.5, 3963.8, 4520.0, 5217.4, 6077.8, 7056.6, 8086.4, 9120.8, 10146.2, 11172.5, 12245.4, 13381.3, 14578.2,
    15832.9, 17136.9, 18484.1, 19874.0, 21268.5, 22686.2, 24139.9, 25453.7, 26908.4, 28555.0, 30395.7, 32482.5,
    34891.6, 37740.9, 41228.2, 45724.0, 52060.6, 62893.1,
]


def discretization(K: int, N: float = 15000.0) -> np.ndarray:
    """K+1 interval boundaries (last = inf). K == 69 gives the reference's 30-100-2000 grid;
    any other K uses quantiles of the constant-size coalescent."""
    if K == len(DISC_30_100_2000):
        d = np.array(DISC_30_100_2000 + [np.inf], dtype=np.float64)
    else:
        q = np.arange(K, dtype=np.float64) / K
        d = np.concatenate([-N * np.log1p(-q), [np.inf]])
    return d


def genetic_distance_keys(max_gen: float = 0.3) -> np.ndarray:
    """The reference's key grid: 0, then 1e-10 and its successors with 3 significant digits
    (DecodingQuantities.java:81-87 and nextGen at :159-165). 6600 keys for max_gen = 0.3."""
    keys = [0.0]
    g = 1e-10
    while g < max_gen:
        keys.append(g)
        g1e10 = g * 1e10
        log10 = int(max(0, math.floor(math.log10(g1e10)) - 2))
        factor = 10.0 ** log10
        g = (round(g1e10 / factor) + 1) * factor / 1e10
    return np.array(keys, dtype=np.float64)


@dataclass
class ModelTables:
    """What a decoding-quantities file holds (DecodingQuantities.hpp:51-68), as arrays."""

    K: int
    csfs_samples: int
    discretization: np.ndarray  # [K+1] f32
    expected_times: np.ndarray  # [K] f32
    initial_state_prob: np.ndarray  # [K] f32
    column_ratios: np.ndarray  # [K] f32 (zero padded)
    keys: np.ndarray  # [R] f32 genetic-distance keys
    D: np.ndarray  # [R][K] f32
    B: np.ndarray  # [R][K] f32 (last column 0)
    U: np.ndarray  # [R][K] f32 (last column 0)
    RR: np.ndarray  # [R][K] f32 (last column 0)
    classic_emission: np.ndarray  # [2][K]
    compressed_emission: np.ndarray  # [2][K]
    folded_ascertained_csfs: np.ndarray  # [n-1][2][K] (rows above n/2 zero)
    ascertained_csfs: np.ndarray  # [n-1][3][K]
    csfs: np.ndarray  # [n-1][3][K]
    folded_csfs: np.ndarray  # [n-1][2][K]
    time_vector: np.ndarray = field(default_factory=lambda: np.zeros(1, np.float32))
    # sequence mode: emission of a run of d homozygous bases, keyed by roundPhysical(d) (HMM.cpp:762-765)
    homozygous_keys: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))  # [H] int32
    homozygous: np.ndarray = field(default_factory=lambda: np.zeros((0, 0), np.float32))  # [H][K]


def physical_distance_keys(max_bp: int = 2_000_000) -> np.ndarray:
    """Every value asmc::roundPhysical(d, 2) can return for d <= max_bp (HmmUtils.cpp:81-94): 3 significant
    digits, i.e. 1..999, then multiples of 10 up to 9990, of 100 up to 99900, ..."""
    keys = list(range(1, 1000))
    factor = 10
    while keys[-1] < max_bp:
        keys += [m * factor for m in range(100, 1000)]
        factor *= 10
    keys.append(1000 * (factor // 10))
    return np.array(sorted(set(k for k in keys if k <= max(max_bp, 1000))), np.int32)


def make_model_tables(K: int = 69, N: float = 15000.0, mu: float = 1.0e-5, csfs_samples: int = 50,
                      keys: np.ndarray | None = None) -> ModelTables:
    """Constant-size SMC tables (SURVEY.md App. B closed forms), computed in float64 and stored fp32.

    With h_j = P(coalesce in interval j), s_i the expected time in interval i and r the genetic
    distance: T[i][j>i] = g(s_i) h_j, T[i][j<i] = B[j], T[i][i] = D[i] (rows sum to 1), hence
    U[i] = g(s_i) h_{i+1}, columnRatios[j] = h_{j+1}/h_j, RR[i] = g(s_i)/g(s_{i+1}).
    """
    d = discretization(K, N)
    lo, hi = d[:-1], d[1:]
    ea, eb = np.exp(-lo / N), np.exp(-hi / N)  # eb[-1] == 0
    h = ea - eb
    with np.errstate(invalid="ignore"):
        hb = np.where(np.isinf(hi), 0.0, hi * eb)
    s = (lo * ea - hb) / h + N  # E[T | T in interval], exponential(1/N)
    if keys is None:
        keys = genetic_distance_keys()
    keys = np.asarray(keys, dtype=np.float64)
    R = keys.shape[0]
    D = np.zeros((R, K))
    Bv = np.zeros((R, K))
    U = np.zeros((R, K))
    RR = np.zeros((R, K))
    tail = np.concatenate([np.cumsum(h[::-1])[::-1][1:], [0.0]])  # sum_{j>i} h_j
    for ri, r in enumerate(keys):
        if r == 0.0:
            D[ri, :] = 1.0
            RR[ri, : K - 1] = 1.0
            continue
        a = 1.0 / N - 2.0 * r
        g = 2.0 * r * np.expm1(a * s) / a
        e2a, e2b = np.exp(-2.0 * r * lo), np.exp(-2.0 * r * hi)
        b = (2.0 * r / (N * a)) * ((e2a - e2b) / (2.0 * r) - N * h)
        b = np.maximum(b, 0.0)
        up = g * tail
        below = np.concatenate([[0.0], np.cumsum(b)[:-1]])  # sum_{j<i} B[j]
        tot = below + up
        # keep every row a probability vector even at the largest distances of the grid
        scale = np.where(tot > 0.999, 0.999 / np.maximum(tot, 1e-300), 1.0)
        if np.any(scale < 1.0):
            sc = scale.min()
            g = g * sc
            b = b * sc
            up = g * tail
            below = np.concatenate([[0.0], np.cumsum(b)[:-1]])
        D[ri] = 1.0 - below - up
        Bv[ri, : K - 1] = b[: K - 1]
        U[ri, : K - 1] = g[: K - 1] * h[1:]
        RR[ri, : K - 1] = g[: K - 1] / g[1:]
    col = np.zeros(K)
    col[: K - 1] = h[1:] / h[:-1]
    classic = np.stack([np.exp(-2.0 * mu * s), -np.expm1(-2.0 * mu * s)])
    # Ascertained-array style emissions: heterozygous probability grows with coalescence time;
    # a smooth dependence on the undistinguished count u keeps every CSFS row distinct.
    n = csfs_samples
    fac = np.zeros((n - 1, 2, K))
    for u in range(n // 2 + 1):
        w = (u + 1.0) / (n // 2 + 1.0)
        het = -np.expm1(-2.0 * mu * s) * (0.4 + 0.6 * w) + 1e-6
        fac[u, 1] = het
        fac[u, 0] = (1.0 - het) * (1.0 - 0.25 * w)
    asc = np.zeros((n - 1, 3, K))
    csfs = np.zeros((n - 1, 3, K))
    for u in range(n - 1):
        w = (u + 1.0) / n
        het = -np.expm1(-2.0 * mu * s) * (0.4 + 0.6 * w) + 1e-6
        asc[u, 1] = het
        asc[u, 0] = (1.0 - het) * (1.0 - 0.25 * w)
        asc[u, 2] = (1.0 - het) * 0.25 * w + 1e-6
        csfs[u] = asc[u] * (0.9 + 0.1 * w)
    fcsfs = fac * 0.97
    hkeys = physical_distance_keys()
    mu_bp = 1.65e-8  # per-base rate for the homozygous stretches between sequence sites
    hom = np.exp(-2.0 * mu_bp * hkeys[:, None].astype(np.float64) * s[None, :])
    f32 = lambda x: np.ascontiguousarray(x, dtype=np.float32)  # noqa: E731
    return ModelTables(
        K=K, csfs_samples=n, discretization=f32(d), expected_times=f32(s), initial_state_prob=f32(h),
        column_ratios=f32(col), keys=f32(keys), D=f32(D), B=f32(Bv), U=f32(U), RR=f32(RR),
        classic_emission=f32(classic), compressed_emission=f32(classic * np.array([[0.98], [1.0]]) + 1e-6),
        folded_ascertained_csfs=f32(fac), ascertained_csfs=f32(asc), csfs=f32(csfs), folded_csfs=f32(fcsfs),
        time_vector=f32(np.array([0.0])), homozygous_keys=hkeys, homozygous=f32(hom),
    )


def _fmt_row(v: np.ndarray) -> str:
    return "\t".join(repr(float(x)) if np.isfinite(x) else "Infinity" for x in v)


def write_decoding_quantities(path: str, t: ModelTables) -> None:
    """Write the gzipped text format parsed by DecodingQuantities.cpp:60-345.
    Values are written with enough digits (repr of the fp32 value as double) to round-trip."""
    K = t.K
    with gzip.open(path, "wt") as f:
        f.write("TransitionType\nSMC\n\n")
        f.write(f"States\n{K}\n\n")
        f.write(f"CSFSSamples\n{t.csfs_samples}\n\n")
        f.write("TimeVector\n" + _fmt_row(t.time_vector) + "\n\n")
        f.write("SizeVector\n15000.0\n\n")
        f.write("Discretization\n" + _fmt_row(t.discretization) + "\n\n")
        f.write("ExpectedTimes\n" + _fmt_row(t.expected_times) + "\n\n")
        for u in range(t.csfs.shape[0]):
            f.write(f"CSFS\t{u}\n" + "\n".join(_fmt_row(t.csfs[u, d]) for d in range(3)) + "\n")
        f.write("\n")
        for u in range(t.folded_csfs.shape[0]):
            f.write(f"FoldedCSFS\t{u}\n" + "\n".join(_fmt_row(t.folded_csfs[u, d]) for d in range(2)) + "\n")
        f.write("\nClassicEmission\n" + "\n".join(_fmt_row(t.classic_emission[d]) for d in range(2)) + "\n\n")
        for u in range(t.ascertained_csfs.shape[0]):
            f.write(f"AscertainedCSFS\t{u}\n" + "\n".join(_fmt_row(t.ascertained_csfs[u, d]) for d in range(3)) + "\n")
        f.write("\n")
        for u in range(t.csfs_samples // 2 + 1):
            f.write(f"FoldedAscertainedCSFS\t{u}\n"
                    + "\n".join(_fmt_row(t.folded_ascertained_csfs[u, d]) for d in range(2)) + "\n")
        f.write("\nCompressedAscertainedEmission\n"
                + "\n".join(_fmt_row(t.compressed_emission[d]) for d in range(2)) + "\n\n")
        if t.homozygous_keys.size:
            f.write("HomozygousEmissions\n")
            for i, key in enumerate(t.homozygous_keys):
                f.write(f"{int(key)}\t" + _fmt_row(t.homozygous[i]) + "\n")
            f.write("\n")
        f.write("initialStateProb\n" + _fmt_row(t.initial_state_prob) + "\n\n")
        f.write("ColumnRatios\n" + _fmt_row(t.column_ratios[: K - 1]) + "\n\n")
        for name, tab, ncol in (("RowRatios", t.RR, K - 1), ("Uvectors", t.U, K - 1), ("Bvectors", t.B, K - 1),
                                ("Dvectors", t.D, K)):
            f.write(name + "\n")
            for ri in range(t.keys.shape[0]):
                f.write(repr(float(t.keys[ri])) + "\t" + _fmt_row(tab[ri, :ncol]) + "\n")
            f.write("\n")


@dataclass
class SynthHaps:
    """Haplotypes as raw (unfolded) alleles plus map information."""

    alleles: np.ndarray  # [n_hap][S] uint8, 0/1; haplotype 2i, 2i+1 belong to individual i
    bp: np.ndarray  # [S] int64, strictly increasing
    cm: np.ndarray  # [S] float64 centimorgans


def make_haps(n_hap: int, S: int, seed: int = 1234, n_founders: int = 24, cm_per_mb: float = 1.0,
              bp_per_site: int = 300, switch_per_cm: float = 0.2, noise: float = 2e-3) -> SynthHaps:
    """Founder-mosaic haplotypes: site frequencies from a 1/x spectrum with
    MAF >= 1 %; each haplotype copies founders with switch rate proportional to genetic
    distance, so long shared (IBD-like) stretches exist; sparse noise breaks exact identity."""
    assert n_hap % 2 == 0
    rng = np.random.default_rng(seed)
    bp = np.sort(rng.choice(np.arange(1, S * bp_per_site + 1), size=S, replace=False)).astype(np.int64)
    cm = bp.astype(np.float64) * (cm_per_mb / 1.0e6)
    # founder alleles at frequency f ~ 1/x on [0.01, 0.5]
    f = 0.01 * (50.0 ** rng.random(S))
    founders = (rng.random((n_founders, S)) < f[None, :]).astype(np.uint8)
    # guarantee polymorphism among founders where possible (array-like ascertainment)
    mono = founders.sum(axis=0) == 0
    founders[rng.integers(0, n_founders, size=int(mono.sum())), np.nonzero(mono)[0]] = 1
    dcm = np.diff(cm, prepend=cm[0])
    alleles = np.empty((n_hap, S), dtype=np.uint8)
    for hidx in range(n_hap):
        sw = rng.random(S) < (1.0 - np.exp(-switch_per_cm * dcm))
        sw[0] = True
        seg_id = np.cumsum(sw) - 1
        src = rng.integers(0, n_founders, size=int(seg_id[-1]) + 1)
        alleles[hidx] = founders[src[seg_id], np.arange(S)]
    if noise > 0:
        alleles ^= (rng.random((n_hap, S)) < noise).astype(np.uint8)
    return SynthHaps(alleles=alleles, bp=bp, cm=cm)


def make_haps_blocked(n_hap: int, S: int, seed: int = 1234, n_founders: int = 24, cm_per_mb: float = 1.0,
                      bp_per_site: int = 300, switch_per_cm: float = 0.2, noise: float = 2e-3,
                      block: int = 128) -> SynthHaps:
    """The same founder-mosaic model as ``make_haps`` for big cohorts (10 000 haplotypes x 100 000 sites in about
    half a minute instead of two): haplotypes are drawn a block at a time with array operations.  The random stream
    is consumed in a different order, so the data differ from ``make_haps`` for the same seed -- tests and the
    one-GPU bench keep ``make_haps``; the multi-GPU bench cohort uses this one."""
    assert n_hap % 2 == 0
    rng = np.random.default_rng(seed)
    bp = np.sort(rng.choice(np.arange(1, S * bp_per_site + 1), size=S, replace=False)).astype(np.int64)
    cm = bp.astype(np.float64) * (cm_per_mb / 1.0e6)
    f = 0.01 * (50.0 ** rng.random(S))
    founders = (rng.random((n_founders, S)) < f[None, :]).astype(np.uint8)
    mono = founders.sum(axis=0) == 0
    founders[rng.integers(0, n_founders, size=int(mono.sum())), np.nonzero(mono)[0]] = 1
    dcm = np.diff(cm, prepend=cm[0])
    p_switch = (1.0 - np.exp(-switch_per_cm * dcm)).astype(np.float32)
    alleles = np.empty((n_hap, S), dtype=np.uint8)
    cols = np.arange(S)
    for h0 in range(0, n_hap, block):
        nb = min(block, n_hap - h0)
        sw = rng.random((nb, S), dtype=np.float32) < p_switch[None, :]
        sw[:, 0] = True
        # the founder of a segment is drawn at its first site and carried along the segment
        draw = rng.integers(0, n_founders, size=(nb, S), dtype=np.uint8)
        first = np.where(sw, cols[None, :], 0)
        np.maximum.accumulate(first, axis=1, out=first)
        src = np.take_along_axis(draw, first, axis=1)
        alleles[h0:h0 + nb] = founders[src, cols[None, :]]
        if noise > 0:
            alleles[h0:h0 + nb] ^= (rng.random((nb, S), dtype=np.float32) < noise).astype(np.uint8)
    return SynthHaps(alleles=alleles, bp=bp, cm=cm)


def write_haps_files(root: str, h: SynthHaps, chrom: int = 1, fastsmc_map: bool = True) -> None:
    """Write ``root.hap.gz``, ``root.samples`` and ``root.map`` in the formats the reference reads
    (Data.cpp:397-521 haps, :212-248 samples, :98-141 FastSMC 3-column map / :162-210 plink map)."""
    n_hap, S = h.alleles.shape
    with gzip.open(root + ".hap.gz", "wt") as f:
        for s in range(S):
            f.write(f"{chrom}:{int(h.bp[s])}_1_2 SNP_{int(h.bp[s])} {int(h.bp[s])} 1 2 "
                    + " ".join("1" if a else "0" for a in h.alleles[:, s]) + "\n")
    with open(root + ".samples", "w") as f:
        f.write("ID_1 ID_2 missing\n0 0 0\n")
        for i in range(n_hap // 2):
            f.write(f"1_{i + 1} 1_{i + 1} 0\n")
    if fastsmc_map:
        with open(root + ".map", "w") as f:
            for s in range(S):
                f.write(f"{int(h.bp[s])}\t{cm_rate_placeholder(h, s)!r}\t{float(h.cm[s])!r}\n")
    else:
        with open(root + ".map", "w") as f:
            for s in range(S):
                f.write(f"{chrom}\tSNP_{int(h.bp[s])}\t{float(h.cm[s])!r}\t{int(h.bp[s])}\n")


def write_haps_files_fast(root: str, h: SynthHaps, chrom: int = 1, block: int = 512) -> None:
    """``write_haps_files`` (FastSMC-mode map) for big cohorts: the allele text is built with array operations, a block
    of sites at a time, and every block is its own gzip member (a multi-member stream is a valid .gz) -- seconds
    instead of minutes for 16 384 haplotypes x 20 000 sites."""
    import zlib

    n_hap, S = h.alleles.shape
    with open(root + ".hap.gz", "wb") as f:
        for s0 in range(0, S, block):
            n = min(block, S - s0)
            body = np.empty((n, 2 * n_hap + 1), np.uint8)
            body[:, 0::2] = 32
            body[:, 1:-1:2] = h.alleles[:, s0:s0 + n].T + 48
            body[:, -1] = 10
            out = bytearray()
            for i in range(n):
                b = int(h.bp[s0 + i])
                out += f"{chrom}:{b}_1_2 SNP_{b} {b} 1 2".encode()
                out += body[i].tobytes()
            co = zlib.compressobj(1, zlib.DEFLATED, 31)
            f.write(co.compress(bytes(out)) + co.flush())
    with open(root + ".samples", "w") as f:
        f.write("ID_1 ID_2 missing\n0 0 0\n")
        for i in range(n_hap // 2):
            f.write(f"1_{i + 1} 1_{i + 1} 0\n")
    with open(root + ".map", "w") as f:
        for s in range(S):
            f.write(f"{int(h.bp[s])}\t0.0\t{float(h.cm[s])!r}\n")


def cm_rate_placeholder(h: SynthHaps, s: int) -> float:
    """Second column of the FastSMC map (cM/Mb rate); read and ignored by Data.cpp:116-128."""
    if s == 0:
        return 0.0
    return float((h.cm[s] - h.cm[s - 1]) / max(1, int(h.bp[s] - h.bp[s - 1])) * 1.0e6)


def fold_and_pack(alleles: np.ndarray) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Fold to minor alleles (Data.cpp:462-463, 505-509) and bit-pack.

    Returns (bits[n_hap][ceil(S/64)] uint64 with site s at bit s%64 of word s//64,
             derived_counts[S] int32 (minor count), flipped[S] bool)."""
    n_hap, S = alleles.shape
    cnt = alleles.sum(axis=0, dtype=np.int64)
    minor_is_one = cnt <= n_hap - cnt
    folded = np.where(minor_is_one[None, :], alleles, 1 - alleles).astype(np.uint8)
    derived = np.minimum(cnt, n_hap - cnt).astype(np.int32)
    return pack_bits(folded), derived, ~minor_is_one


def pack_bits(folded: np.ndarray) -> np.ndarray:
    n_hap, S = folded.shape
    W = (S + 63) // 64
    padded = np.zeros((n_hap, W * 64), dtype=np.uint8)
    padded[:, :S] = folded
    return np.packbits(padded, axis=1, bitorder="little").view("<u8").reshape(n_hap, W).copy()
