#!/usr/bin/env python3
"""Runs on the GPU box: HMM.decodeAll of job 3 of 10 on files of the C1 shape (4485 pairs x 6760 sites) with the per-pair
posterior-mean and MAP files on (HMM.hpp:287, 293): what the text output of 30 million numbers a file costs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, os, tempfile, gzip, copy, numpy as np
from fastsmc_amd import api, synth
from oracle import oracle as O
tables = synth.make_model_tables(69); haps = synth.make_haps(300, 6760, seed=1234)
with tempfile.TemporaryDirectory() as d:
    root = os.path.join(d, "syn"); synth.write_haps_files(root, haps, fastsmc_map=False)
    gen = (haps.cm / 100.0).astype(np.float32)
    gen32 = np.array([np.float32(np.float32(c) / np.float32(100.0)) for c in haps.cm], np.float32)
    used = np.unique(np.concatenate([[0.0], O.step_rows(tables.keys, gen)[1][1:], O.step_rows(tables.keys, gen32)[1][1:]]))
    t = copy.copy(tables); sel = np.nonzero(np.isin(t.keys, used.astype(np.float32)))[0]
    t.keys, t.D, t.B, t.U, t.RR = t.keys[sel], t.D[sel], t.B[sel], t.U[sel], t.RR[sel]
    synth.write_decoding_quantities(root + ".decodingQuantities.gz", t)
    p = api.DecodingParams(root, root + ".decodingQuantities.gz")
    p.useKnownSeed = True; p.outFileRoot = os.path.join(d, "out"); p.jobs, p.jobInd = 10, 3
    data = api.Data(p); hmm = api.HMM(data, p)
    hmm.setWritePerPairPosteriorMean(True); hmm.setWritePerPairMap(True)
    for rep in range(2):
        t0 = time.perf_counter(); hmm.decodeAll(p.jobs, p.jobInd); dt = time.perf_counter() - t0
        txt = gzip.open(p.outFileRoot + ".perPairPosteriorMeans.gz", "rt").read()
        print("per-pair mean + MAP files of job 3/10 (C1 shape): decodeAll %.2f s; %d numbers a file, %.0f MB of text" % (dt, txt.count(" ") + txt.count("\n") + 1, len(txt) / 1e6), flush=True)
