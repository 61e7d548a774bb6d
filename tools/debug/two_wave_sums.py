import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fastsmc_amd import capi, synth
from oracle import oracle as O

def problem(K, S=200, seed=11):
    tables = synth.make_model_tables(K)
    haps = synth.make_haps(64, S, seed=seed, cm_per_mb=25.0, switch_per_cm=0.6)
    bits, derived, flipped = synth.fold_and_pack(haps.alleles)
    pm = O.prepare_model(tables, (haps.cm / 100.0).astype(np.float32), haps.bp, derived, 64, time=200)
    return pm, bits

for K in [int(x) for x in sys.argv[1:]] or (81, 96, 99, 105, 112, 113, 120):
    pm, bits = problem(K)
    pairs = np.array(O.enumerate_all_pairs(32)[:64], np.uint32).view(capi.PAIR_DTYPE).reshape(-1)
    out = {}
    for mode in (0, 1):
        ctx = capi.Context(0)
        ctx.set_two_wave_windows(mode)
        model = ctx.create_model(pm)
        ctx.upload_haps(bits, pm.S)
        ctx.upload_worklist(pairs, capi.whole_sequence_groups(64, pm.S))
        runs = [ctx.decode_sums(model)[0].copy() for _ in range(3)]
        out[mode] = runs
        print(K, "mode", mode, "waves", ctx.last_waves_per_window(), "member", ctx.last_kernel(), "repeatable", all(np.array_equal(runs[0], r) for r in runs))
        ctx.close()
    bad = np.argwhere(out[0][0] != out[1][0])
    print(K, "mismatches", len(bad), "sites", (bad[:, 0].min(), bad[:, 0].max()) if len(bad) else None, "states", (bad[:, 1].min(), bad[:, 1].max()) if len(bad) else None, "mid", pm.S // 2)
