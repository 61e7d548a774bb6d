import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fastsmc_amd import capi, synth
from oracle import oracle as O

K = 112
tables = synth.make_model_tables(K)
haps = synth.make_haps(64, 200, seed=11, cm_per_mb=25.0, switch_per_cm=0.6)
bits, derived, flipped = synth.fold_and_pack(haps.alleles)
pm = O.prepare_model(tables, (haps.cm / 100.0).astype(np.float32), haps.bp, derived, 64, time=200)
allp = np.array(O.enumerate_all_pairs(32)[:64], np.uint32)
ctx = capi.Context(0)
model = ctx.create_model(pm)
ctx.upload_haps(bits, pm.S)
for n in (1, 2, 17, 64):
    pairs = allp[:n].view(capi.PAIR_DTYPE).reshape(-1)
    ctx.upload_worklist(pairs, capi.whole_sequence_groups(n, pm.S))
    ctx.set_two_wave_windows(1)
    post = ctx.decode_posteriors(model)[0]
    ctx.set_two_wave_windows(0)
    s2 = ctx.decode_sums(model)[0]
    seq = np.zeros((pm.S, pm.K), np.float32)
    for v in range(n):
        seq = seq + post[:, :, v]
    bad = s2 != seq
    print("pairs", n, "waves", ctx.last_waves_per_window(), "bad", int(bad.sum()), "of", bad.size)
    if n == 1 and bad.any():
        # what IS in a bad entry?  look for the value among this site's posteriors (any state) and the neighbours'
        for (s, k) in np.argwhere(bad)[:12]:
            val = s2[s, k]
            hit = [(ds, int(kk)) for ds in (-2, -1, 0, 1, 2) if 0 <= s + ds < pm.S for kk in np.argwhere(post[s + ds, :, 0] == val).ravel()]
            print("  site", s, "state", k, "got", val, "want", seq[s, k], "found at (dsite, state):", hit[:4])
        print("  bad states at site 0:", np.argwhere(bad[0]).ravel()[:40], " bad sites for state 0:", np.argwhere(bad[:, 0]).ravel()[:40])
ctx.close()
