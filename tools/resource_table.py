#!/usr/bin/env python3
"""Per-kernel resource table of the HIP library as the compiler reports it (-Rpass-analysis=kernel-resource-usage): VGPRs,
AGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, LDS bytes per workgroup, waves per SIMD -- one row per
instantiation of every family member, stamped with the hash of the sources they were compiled from.
Usage: tools/resource_table.py [out.json]   (default profiles/r05_kernel_resources.json; no GPU needed)"""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastsmc_amd.build import (EXACT_MEMBERS, HIPCC_FLAGS, KT_MEMBERS, W2_MEMBERS, exact_define,  # noqa: E402
                               hip_source_hash)

FIELDS = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
          "Occupancy [waves/SIMD]": "waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
          "LDS Size [bytes/block]": "lds_bytes_per_workgroup", "TotalSGPRs": "sgprs"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.strip().split("\n")


def member(define):
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + exact_define()
    r = subprocess.run(["hipcc", *flags, *define.split(), "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        "-Wno-unused-command-line-argument", "-o", "/dev/null",
                        os.path.join(ROOT, "fastsmc_amd", "csrc", "fsmc_inst.hip")], capture_output=True, text=True)
    rows, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = {"mangled": m.group(1)}
            rows.append(cur)
            continue
        for k, v in FIELDS.items():
            m = re.search(re.escape(k) + r": (\d+)", ln)
            if m and cur is not None:
                cur[v] = int(m.group(1))
    return rows


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05_kernel_resources.json")
    # every member the build compiles (fastsmc_amd/build.py): padded and exact lane-per-pair members, wave-group members
    defs = ([f"-DFSMC_INSTANCE_KT={k}" for k in KT_MEMBERS + EXACT_MEMBERS]
            + [f"-DFSMC_INSTANCE_W2={kh} -DFSMC_INSTANCE_NW={nw}" for kh, nw in W2_MEMBERS])
    with ThreadPoolExecutor(max_workers=min(len(defs), os.cpu_count() or 1)) as ex:
        rows = [r for rs in ex.map(member, defs) for r in rs]
    for r, name in zip(rows, demangle([r["mangled"] for r in rows])):
        r["kernel"] = re.sub(r"\(fsmc::KParams\)$", "", name).replace("void fsmc::", "")
        del r["mangled"]
    rows.sort(key=lambda r: r["kernel"])
    doc = {"lib_hash": hip_source_hash(), "flags": " ".join(HIPCC_FLAGS),
           "template_arguments": {"decode_kernel": "<KT, MODE (0 IBD, 1 dump, 2 per pair, 3 sums), TRACK, SEQ, HALF, DUAL>",
                                  "decode_kernel_w2": "<KH, MODE, TRACK, SEQ, NW>"},
           "kernels": rows}
    json.dump(doc, open(out, "w"), indent=1)
    print(f"{len(rows)} kernels -> {out}")
    for r in rows:
        if "kModeIbd" in r["kernel"] or ", 0," in r["kernel"] or "(fsmc::Mode)0" in r["kernel"]:
            print(r["kernel"], {k: r.get(k) for k in ("vgprs", "vgpr_spills", "sgpr_spills", "scratch_bytes_per_lane",
                                                       "waves_per_simd", "lds_bytes_per_workgroup")})


if __name__ == "__main__":
    main()
