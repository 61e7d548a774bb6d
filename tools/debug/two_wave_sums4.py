import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fastsmc_amd import capi, synth
from oracle import oracle as O

K = 112
tables = synth.make_model_tables(K)
haps = synth.make_haps(64, 200, seed=11, cm_per_mb=25.0, switch_per_cm=0.6)
bits, derived, flipped = synth.fold_and_pack(haps.alleles)
folded = np.where(flipped[None, :], 1 - haps.alleles, haps.alleles).astype(np.uint8)
pm = O.prepare_model(tables, (haps.cm / 100.0).astype(np.float32), haps.bp, derived, 64, time=200)
pairs_l = O.enumerate_all_pairs(32)[:64]
allp = np.array(pairs_l, np.uint32)
ob = np.stack([folded[a] ^ folded[b] for a, b in pairs_l])
hb = np.stack([folded[a] & folded[b] for a, b in pairs_l])
post, beta, afwd = O.decode_batch(pm, ob, hb, 0, pm.S, want_alpha_fwd=True)
ctx = capi.Context(0)
model = ctx.create_model(pm)
ctx.upload_haps(bits, pm.S)
for n in (1, 64):
    ctx.upload_worklist(allp[:n].view(capi.PAIR_DTYPE).reshape(-1), capi.whole_sequence_groups(n, pm.S))
    ctx.set_two_wave_windows(0)
    s2 = ctx.decode_sums(model)[0]
    want = np.zeros((pm.S, pm.K), np.float32)
    mid = pm.S // 2
    for v in range(n):
        want[mid:] = want[mid:] + afwd[mid:, :, v]
        want[:mid] = want[:mid] + beta[:mid, :, v]
    bad = s2 != want
    print("pairs", n, "waves", ctx.last_waves_per_window(), "bad", int(bad.sum()))
    if bad.any():
        sites = np.unique(np.argwhere(bad)[:, 0])
        print("  bad sites:", sites[:60])
        s0 = sites[sites >= mid].min() if (sites >= mid).any() else None
        if s0 is not None:
            print("  first bad alpha site", s0, "bad states", np.argwhere(bad[s0]).ravel()[:30])
            print("   got ", s2[s0, :6], "\n   want", want[s0, :6])
        sb = sites[sites < mid].max() if (sites < mid).any() else None
        if sb is not None:
            print("  first bad beta site (descending)", sb, "bad states", np.argwhere(bad[sb]).ravel()[:30])
ctx.close()
