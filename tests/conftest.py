import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
