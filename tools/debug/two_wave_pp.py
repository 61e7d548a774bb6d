import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fastsmc_amd import capi, synth
from oracle import oracle as O

K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tables = synth.make_model_tables(K)
haps = synth.make_haps(64, 150, seed=3, cm_per_mb=25.0, switch_per_cm=0.6)
bits, derived, flipped = synth.fold_and_pack(haps.alleles)
pm = O.prepare_model(tables, (haps.cm / 100.0).astype(np.float32), haps.bp, derived, 64, time=200)
pairs = np.array(O.enumerate_all_pairs(32)[:151], np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
ctx = capi.Context(0)
model = ctx.create_model(pm)
ctx.upload_haps(bits, pm.S)
ctx.upload_worklist(pairs, capi.whole_sequence_groups(151, pm.S))
ctx.set_two_wave_windows(1)
ref_mean, ref_map = ctx.decode_per_pair(model, pm.exp_times)
ctx.set_two_wave_windows(0)
for it in range(5):
    mean, mp = ctx.decode_per_pair(model, pm.exp_times)
    bad = np.argwhere(mean != ref_mean)
    print("run", it, "waves", ctx.last_waves_per_window(), "bad", len(bad), "map bad", int((mp != ref_map).sum()))
    if len(bad):
        print("   pairs", np.unique(bad[:, 0])[:20], "... n pairs", len(np.unique(bad[:, 0])))
        print("   sites", np.unique(bad[:, 1])[:60])
ctx.close()
