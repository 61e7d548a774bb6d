#!/usr/bin/env python3
"""Static check of the compiled decode kernels: the hand-placed scalar loads (inline asm) return their data some
hundred cycles after they issue, and the compiler does not know that.  Between such a load and the next
`s_waitcnt lgkmcnt(0)` no instruction may read or write the load's destination SGPRs (a spill would save garbage;
a reuse would be overwritten when the load lands -- a clobbered pointer is a GPU memory fault).
Reads /tmp/fsmc_isa.s (tools/isa_stats.py writes it); exits 1 on a violation.  Run by tests/test_isa_hazards.py."""
import re
import sys


def sgprs(tok: str):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    return set()


def all_sgprs(line: str):
    """SGPRs an instruction names.  A packed VALU instruction whose op_sel_hi bit for an SGPR-pair source is 0 (and
    whose op_sel bit is not set) reads the pair's LOW register for both halves: the high register is not touched."""
    out = set()
    low_only = set()
    m = re.search(r"op_sel_hi:\[([01,]+)\]", line)
    if line.startswith("v_pk_") and m and "op_sel:" not in line:
        hi = [int(x) for x in m.group(1).split(",")]
        ops = [t.strip() for t in line.split(None, 1)[1].split("op_sel_hi")[0].split(",")]
        for idx, tok in enumerate(ops[1:]):  # sources
            if idx < len(hi) and hi[idx] == 0 and re.fullmatch(r"s\[(\d+):(\d+)\]", tok):
                low_only.add(tok)
    for t in re.findall(r"s\[\d+:\d+\]|\bs\d+\b", line):
        regs = sgprs(t)
        if t in low_only:
            regs = {min(regs)}
        out |= regs
    return out


def check(path: str) -> int:
    txt = open(path).read()
    parts = re.split(r"\n(_ZN4fsmc\d+decode_kernel\w+):[^\n]*\n", txt)
    bad = 0
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
        inflight = {}  # sgpr -> line of the load
        in_asm = False
        for ln_no, ln in enumerate(body.split("\n")):
            s = ln.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith((";", ".")) or s.endswith(":"):
                # a label: control flow may merge here; keep tracking conservatively (straight-line order)
                continue
            op = s.split()[0]
            if op == "s_waitcnt" and "lgkmcnt(0)" in s:
                inflight.clear()
                continue
            if op.startswith("s_load_dword") and in_asm:
                toks = s.split()
                dst = sgprs(toks[1])
                addr = sgprs(toks[2])
                # the address pair must not be the destination of a load that has not returned
                if addr & set(inflight):
                    print(f"{name}: line {ln_no}: address s{sorted(addr & set(inflight))} is the destination of a "
                          f"load in flight: {s}")
                    bad += 1
                clash = dst & set(inflight)
                if clash:
                    print(f"{name}: line {ln_no}: load into s{sorted(clash)} while an earlier load to it is in flight: {s}")
                    bad += 1
                for r in dst:
                    inflight[r] = ln_no
                continue
            if inflight:
                used = all_sgprs(s)
                clash = used & set(inflight)
                if clash:
                    print(f"{name}: line {ln_no}: touches s{sorted(clash)} (loaded at line "
                          f"{min(inflight[r] for r in clash)}) before the wait: {s}")
                    bad += 1
    return bad


if __name__ == "__main__":
    n = check(sys.argv[1] if len(sys.argv) > 1 else "/tmp/fsmc_isa.s")
    print(f"{n} violation(s)")
    sys.exit(1 if n else 0)
