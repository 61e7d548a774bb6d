#!/usr/bin/env python3
"""Regenerates tests/golden/small_problem_oracle.npz from the CPU oracle (oracle/hmm_oracle.c).

These are NOT reference outputs (the reference cannot be built or run in this environment and its own golden
files need missing model blobs); they freeze the oracle's results for the seeded synthetic
problem of tests/conftest.py so that (a) an accidental change of the oracle is caught on CPU and (b) the GPU
path is also compared with committed numbers.  Run from the repo root: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from fastsmc_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def build():
    tables = synth.make_model_tables(69)
    haps = synth.make_haps(64, 640, seed=7, cm_per_mb=25.0, switch_per_cm=0.6)
    _, derived, flipped = synth.fold_and_pack(haps.alleles)
    folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
    gen = (haps.cm / 100.0).astype(np.float32)
    pm = O.prepare_model(tables, gen, haps.bp, derived, 64, time=50)
    pairs = O.enumerate_all_pairs(32)[:192]
    recs = O.decode_pairs_ibd(pm, folded, pairs, batch_size=64)
    sub = pairs[:8]
    ob = np.stack([folded[a] ^ folded[b] for a, b in sub])
    hb = np.stack([folded[a] & folded[b] for a, b in sub])
    post, _ = O.decode_batch(pm, ob, hb, 0, pm.S)
    mean, mp, _ = O.per_pair_output(pm, post, 8)
    return dict(records=recs, posterior_first8=post.astype(np.float32), mean_first8=mean, map_first8=mp,
                pairs=np.array(pairs, np.uint32), e1_checksum=np.float64(pm.e1.astype(np.float64).sum()),
                state_threshold=np.int32(pm.state_threshold), probability_threshold=np.float32(pm.probability_threshold))


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_problem_oracle.npz")
    np.savez_compressed(out, **build())
    print("wrote", out, os.path.getsize(out), "bytes")
