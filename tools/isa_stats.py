#!/usr/bin/env python3
"""Static instruction mix of the decode kernels: compiles one family member (csrc/fsmc_inst.hip, default
-DFSMC_INSTANCE_KT=69; set ISA_MEMBER=kt<n> or w2_<n>) to gfx950 assembly (device only) and counts instructions per
kernel instantiation.  Usage: tools/isa_stats.py [filter] [-- extra hipcc flags]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fastsmc_amd.build import HIPCC_FLAGS  # noqa: E402

args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    extra = args[i + 1:]
    args = args[:i]
flt = args[0] if args else "ILi69ELi0E"
out = "/tmp/fsmc_isa.s"
flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
member = os.environ.get("ISA_MEMBER", "kt69")
define = "-DFSMC_INSTANCE_W2=" + member[3:] if member.startswith("w2_") else "-DFSMC_INSTANCE_KT=" + member[2:]
subprocess.run(["hipcc", *flags, *extra, define, "-S", "--cuda-device-only", "-Wno-unused-command-line-argument", "-o",
                out, os.path.join(ROOT, "fastsmc_amd/csrc/fsmc_inst.hip")], check=True)
txt = open(out).read()
parts = re.split(r"\n(_ZN4fsmc\d+decode_kernel\w+):[^\n]*\n", txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
    if flt not in name:
        continue
    ins = [ln.strip().split()[0] for ln in body.split("\n")
           if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    grp = lambda pre: sum(v for k, v in c.items() if k.startswith(pre))  # noqa: E731
    print(f"{name}\n  total {len(ins)}  valu {grp('v_')}  v_pk {grp('v_pk')}  v_mov {grp('v_mov')}  "
          f"scratch {sum(v for k, v in c.items() if 'scratch' in k)}  s_load {grp('s_load')}  ds {grp('ds_')}  "
          f"global {grp('global_')}  s_waitcnt {c['s_waitcnt']}")
    if os.environ.get("ISA_TOP"):
        print("  ", c.most_common(25))
