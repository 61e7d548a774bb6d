#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in /tmp/fsmc_isa.s (written by tools/isa_stats.py).
Usage: tools/isa_blocks.py <mangled-name-substring> [min_instrs]"""
import collections
import re
import sys

flt = sys.argv[1]
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 120
txt = open("/tmp/fsmc_isa.s").read()
parts = re.split(r"\n(_ZN4fsmc\d+decode_kernel\w+):[^\n]*\n", txt)
for i in range(1, len(parts), 2):
    if flt not in parts[i]:
        continue
    body = parts[i + 1].split(".Lfunc_end")[0]
    lines = body.split("\n")
    blocks = []
    cur = None
    for j, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", ln)
        if m:
            cur = [m.group(1), j, collections.Counter()]
            blocks.append(cur)
        elif cur is not None and ln.startswith("\t") and not ln.strip().startswith((".", ";")):
            cur[2][ln.strip().split()[0]] += 1
    print(parts[i])
    for name, j, c in blocks:
        n = sum(c.values())
        if n < mn:
            continue
        g = lambda pre: sum(v for k, v in c.items() if k.startswith(pre))  # noqa: E731
        print(f"  {name:12s} line {j:5d} n {n:4d} valu {g('v_'):4d} pk {g('v_pk'):3d} readlane {c['v_readlane_b32']:3d} "
              f"writelane {c['v_writelane_b32']:3d} v_mov {g('v_mov'):3d} scratch {sum(v for k, v in c.items() if 'scratch' in k):3d} "
              f"smem {g('s_load'):3d} salu {sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith(('s_load', 's_waitcnt', 's_nop'))):3d} "
              f"ds {g('ds_'):3d} vmem {g('global_'):3d} wait {c['s_waitcnt']:3d} nop {c['s_nop']:3d}")
