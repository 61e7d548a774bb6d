#!/usr/bin/env python3
"""Builds an experimental variant of the HIP library (same C ABI) into fastsmc_amd/variants/lib<name>.so; run it with
FSMC_HIP_LIB=fastsmc_amd/variants/lib<name>.so python bench.py ...

The units are the build's own (fastsmc_amd/build.py: KT_MEMBERS + EXACT_MEMBERS with exact_define(), W2_MEMBERS), so a
variant's -D switches reach every member the library ships.  `--only w2` / `--only kt` recompile just that family with
the extra flags and take every other object from the shipped build (fastsmc_amd/csrc/obj/: build the library first);
`--capi` recompiles fsmc_capi.hip too (a variant that changes the host side of the launch); `--src-dir` compiles an
edited COPY of csrc/ (the library's own sources, hence its hash, stay as they are).  Every recompiled unit's ISA goes
through the in-flight scalar-load check (tools/check_inflight_sgprs.py) and `--resources <substring>` prints registers /
spills / scratch / LDS of the instantiations whose mangled name contains the substring.

Usage: tools/build_variant.py <name> [--only all|w2|kt] [--capi] [--src-dir DIR] [--resources SUBSTR] [-- hipcc flags]"""
import argparse
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fastsmc_amd.build import (CSRC, EXACT_MEMBERS, HIPCC_FLAGS, KT_MEMBERS, OBJ_DIR, exact_define,  # noqa: E402
                               w2_units)


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--only", choices=("all", "w2", "kt"), default="all")
    ap.add_argument("--capi", action="store_true")
    ap.add_argument("--src-dir", default=CSRC)
    ap.add_argument("--resources", default="")
    ap.add_argument("--units", default="", help="comma-separated unit names (kt112, w2_64x6 ...): only these are recompiled")
    ap.add_argument("--extra-w2", default="", help="wave-group members beyond the shipped ones, '128x4,128x3' (states per "
                    "wave x waves): compiled from --src-dir, whose fsmc_instances.h must list them in FSMC_ALL_W2")
    args = ap.parse_args(argv)
    out_dir = os.path.join(ROOT, "fastsmc_amd", "variants")
    obj = os.path.join(out_dir, "obj_" + args.name)
    os.makedirs(obj, exist_ok=True)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + exact_define() + extra
    inst = os.path.join(args.src_dir, "fsmc_inst.hip")
    kt = [(f"kt{k}", inst, [f"-DFSMC_INSTANCE_KT={k}"]) for k in KT_MEMBERS + EXACT_MEMBERS]
    w2 = [(name, inst, defs) for name, defs in w2_units()]
    extra_w2 = [tuple(int(x) for x in e.split("x")) for e in args.extra_w2.split(",") if e]
    extra_w2 = [(f"w2_{kh}x{nw}", inst, [f"-DFSMC_INSTANCE_W2={kh}", f"-DFSMC_INSTANCE_NW={nw}"]) for kh, nw in extra_w2]
    rest = [("idsort", os.path.join(args.src_dir, "fsmc_identify_sort.hip"), []),
            ("idseeds", os.path.join(args.src_dir, "fsmc_identify_seeds.hip"), [])]
    capi = [("capi", os.path.join(args.src_dir, "fsmc_capi.hip"), [])]
    units = {"all": kt + w2 + rest + capi, "w2": w2, "kt": kt}[args.only]
    if args.capi and args.only != "all":
        units = units + capi
    if args.units:
        want = set(args.units.split(","))
        units = [u for u in kt + w2 + rest + capi if u[0] in want]
    units = units + extra_w2
    names = {u[0] for u in units}

    def compile_unit(u):
        name, src, defs = u
        o = os.path.join(obj, name + ".o")
        r = subprocess.run(["hipcc", *flags, *defs, "-c", "-Rpass-analysis=kernel-resource-usage", "-o", o, src],
                           capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise RuntimeError("hipcc failed for " + name)
        open(os.path.join(obj, name + ".res"), "w").write(r.stderr)
        if src.endswith("fsmc_inst.hip"):
            subprocess.run(["hipcc", *[f for f in flags if f != "-fPIC"], *defs, "-S", "--cuda-device-only",
                            "-Wno-unused-command-line-argument", "-o", os.path.join(obj, name + ".s"), src], check=True,
                           capture_output=True)
        return o

    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_unit, units))
    import check_inflight_sgprs as chk

    bad = 0
    for name, src, _ in units:
        s = os.path.join(obj, name + ".s")
        if os.path.exists(s):
            rc = chk.check(s)
            print("in-flight SGPR check", name, "OK" if rc == 0 else "VIOLATION")
            bad |= rc
    if args.resources:
        for name, _, _ in units:
            txt = open(os.path.join(obj, name + ".res")).read()
            for b in txt.split("Function Name: ")[1:]:
                fn = b.split()[0]
                if args.resources not in fn:
                    continue
                f = lambda key: (re.search(key + r": (\d+)", b) or [None, "?"])[1]  # noqa: E731
                print(fn, "VGPRs", f("VGPRs"), "AGPRs", f("AGPRs"), "spill VGPR", f("VGPRs Spill"), "spill SGPR",
                      f("SGPRs Spill"), "scratch", f(r"ScratchSize \[bytes/lane\]"), "LDS", f(r"LDS Size \[bytes/block\]"),
                      "waves/SIMD", f(r"Occupancy \[waves/SIMD\]"))
    # every unit that was not recompiled is the shipped build's object
    shipped = [os.path.join(OBJ_DIR, f) for f in sorted(os.listdir(OBJ_DIR)) if f.endswith(".o") and f[:-2] not in names]
    lib = os.path.join(out_dir, f"lib{args.name}.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, *shipped], check=True)
    shutil.rmtree(obj)
    print("built", lib)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
