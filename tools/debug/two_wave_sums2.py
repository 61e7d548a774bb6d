import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fastsmc_amd import capi, synth
from oracle import oracle as O

K = int(sys.argv[1]) if len(sys.argv) > 1 else 112
tables = synth.make_model_tables(K)
haps = synth.make_haps(64, 200, seed=11, cm_per_mb=25.0, switch_per_cm=0.6)
bits, derived, flipped = synth.fold_and_pack(haps.alleles)
pm = O.prepare_model(tables, (haps.cm / 100.0).astype(np.float32), haps.bp, derived, 64, time=200)
pairs = np.array(O.enumerate_all_pairs(32)[:64], np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
ctx = capi.Context(0)
model = ctx.create_model(pm)
ctx.upload_haps(bits, pm.S)
ctx.upload_worklist(pairs, capi.whole_sequence_groups(64, pm.S))
ctx.set_two_wave_windows(1)
post1 = ctx.decode_posteriors(model)[0]
s1 = ctx.decode_sums(model)[0]
ctx.set_two_wave_windows(0)
post2 = ctx.decode_posteriors(model)[0]
s2 = ctx.decode_sums(model)[0]
print("waves", ctx.last_waves_per_window(), "posteriors equal", np.array_equal(post1, post2))
seq = np.zeros((pm.S, pm.K), np.float32)
for v in range(64):
    seq = seq + post1[:, :, v]
print("one-wave == sequential", np.array_equal(s1, seq), " two-wave == sequential", np.array_equal(s2, seq))
bad = s2 != seq
print("bad per state block:", [int(bad[:, a:a + 16].sum()) for a in range(0, pm.K, 16)])
print("bad per site block:", [int(bad[a:a + 20].sum()) for a in range(0, pm.S, 20)])
# does a bad entry equal the sum with one pair's value replaced by another site's / state's?
i = np.argwhere(bad)[:5]
for (s, k) in i:
    d = float(s2[s, k]) - float(seq[s, k])
    print("site", s, "state", k, "got", s2[s, k], "want", seq[s, k], "diff", d, "col", post1[s, k, :4])
# candidates: tile row k taken from the neighbouring state / site
for name, cand in (("state+1", np.roll(seq, -1, axis=1)), ("state-1", np.roll(seq, 1, axis=1)), ("site+1", np.roll(seq, -1, axis=0)),
                   ("site-1", np.roll(seq, 1, axis=0))):
    print(name, "explains", int((s2[bad] == cand[bad]).sum()), "of", int(bad.sum()))
ctx.close()
