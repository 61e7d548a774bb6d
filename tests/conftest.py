import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def wave_group_member(K):
    """(waves per group, states per wave) of the wave-group kernel for a model of 128 < K <= 1024 states, as
    fsmc_model_create picks it (csrc/fsmc_capi.hip, w2Member): four waves of 48 / 64 / 80 states, then six, seven, eight
    waves of 64, beyond 512 states eight waves of 80 / 96 / 128."""
    forced = os.environ.get("FSMC_DIAG_W2_MEMBER")  # (A/B runs of another member: the library honours the same variable)
    if forced:
        nw, kh = (int(x) for x in forced.split("x"))
        if nw * kh >= K and ((nw - 1) * kh < K or (kh == 48 and 2 * kh < K)
                             or (nw * kh > 512 and (nw - 2) * kh < K)):  # (the library's condition: w2Member)
            return (nw, kh)
    return ((4, 48) if K <= 192 else (4, 64) if K <= 256 else (4, 80) if K <= 320 else (6, 64) if K <= 384
            else (7, 64) if K <= 448 else (8, 64) if K <= 512 else (8, 80) if K <= 640 else (8, 96) if K <= 768
            else (8, 128))


def expected_member(K):
    """What fsmc_ctx_last_kernel reports: the exact or padded family member for K <= 128; up to 1024 states the
    wave-group kernel (1000 + states per wave with four waves a group, 1000 * waves + states per wave with more);
    beyond, 0 = the any-K kernel (a pair's K-vectors in the workspace)."""
    if K in (69, 50, 100):  # the exact members of the default build (fsmc_instances.h: FSMC_EXACT_KT)
        return K
    if K <= 128:
        return (K + 15) // 16 * 16
    if K > 1024:
        return 0
    nw, kh = wave_group_member(K)
    return 1000 + kh if nw == 4 else 1000 * nw + kh


@pytest.fixture(params=["two-waves-auto", "one-wave"])
def window_waves(request, monkeypatch):
    """The dump / per-pair / sums consumers of a small launch run two waves per window (csrc/fsmc_kernels_bidir.h); the
    tests of a module that asks for this fixture run once that way and once on the one-wave kernels."""
    if request.param == "one-wave":
        monkeypatch.setenv("FSMC_DIAG_TWO_WAVE_WINDOWS", "never")
    return request.param


def build_small_problem():
    """A seeded synthetic problem small enough for the CPU oracle: 64 haplotypes x 640 sites, K = 69."""
    import numpy as np
    from fastsmc_amd import synth
    from oracle import oracle as O

    tables = synth.make_model_tables(69)
    haps = synth.make_haps(64, 640, seed=7, cm_per_mb=25.0, switch_per_cm=0.6)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    gen = (haps.cm / 100.0).astype(np.float32)
    model = O.prepare_model(tables, gen, haps.bp, derived, 64, time=50)
    return dict(tables=tables, haps=haps, bits=bits, folded=folded, gen=gen, model=model)


@pytest.fixture(scope="session")
def small_problem():
    return build_small_problem()


@pytest.fixture(scope="session")
def seq_problem():
    """Sequence-mode (decodingSequence) variant: denser sites with varying spacing, 64 haplotypes x 400 sites,
    folded CSFS / classic emissions, two transition steps per site with a homozygous stretch in between."""
    import numpy as np
    from fastsmc_amd import synth
    from oracle import oracle as O

    tables = synth.make_model_tables(69)
    haps = synth.make_haps(64, 400, seed=21, cm_per_mb=1.2, bp_per_site=2500, switch_per_cm=2.0)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    gen = (haps.cm / 100.0).astype(np.float32)
    model = O.prepare_model(tables, gen, haps.bp, derived, 64, time=50, decoding_sequence=True)
    return dict(tables=tables, haps=haps, bits=bits, folded=folded, gen=gen, model=model)
