#!/usr/bin/env python3
"""Static check of the compiled decode kernels: the hand-placed scalar loads (inline asm) return their data some
hundred cycles after they issue, and the compiler does not know that.  Between such a load and the next
`s_waitcnt lgkmcnt(0)` ON ANY PATH no instruction may read or write the load's destination SGPRs (a spill would save
garbage; a reuse would be overwritten when the load lands -- a clobbered pointer is a GPU memory fault).

The check is a forward data-flow over the kernel's control-flow graph: the set of in-flight destination registers at
the top of a basic block is the union over its predecessors; an `s_waitcnt` whose lgkmcnt is 0 empties it.
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


class Block:
    def __init__(self, name):
        self.name = name
        self.ins = []  # (line number, text, inside inline asm)
        self.succ = []
        self.falls = True


def blocks_of(body: str):
    """Basic blocks: a block starts at a label and after every branch instruction."""
    blocks = [Block("entry")]
    in_asm = False
    n_anon = 0
    for ln_no, ln in enumerate(body.split("\n")):
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks.append(Block(m.group(1)))
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        blocks[-1].ins.append((ln_no, s, in_asm))
        op = s.split()[0]
        if op == "s_branch" or op.startswith("s_cbranch") or op in ("s_endpgm", "s_setpc_b64"):
            n_anon += 1
            blocks.append(Block(f".anon{n_anon}"))
    by_name = {b.name: b for b in blocks}
    for i, b in enumerate(blocks):
        falls = True
        if b.ins:
            s = b.ins[-1][1]
            op = s.split()[0]
            if op == "s_branch":
                b.succ.append(s.split()[1])
                falls = False
            elif op.startswith("s_cbranch"):
                b.succ.append(s.split()[1])
            elif op in ("s_endpgm", "s_setpc_b64"):
                falls = False
        if falls and i + 1 < len(blocks):
            b.succ.append(blocks[i + 1].name)
    return blocks, by_name


def transfer(b: Block, inflight: dict, report, name):
    """Walk a block with the in-flight map {sgpr: line of its load}; returns the map at the block's end."""
    cur = dict(inflight)
    bad = 0
    for ln_no, s, in_asm in b.ins:
        op = s.split()[0]
        if op == "s_waitcnt" and "lgkmcnt(0)" in s:
            cur.clear()
            continue
        if op.startswith("s_load_dword") and in_asm:
            toks = s.split()
            dst = sgprs(toks[1])
            addr = sgprs(toks[2])
            if report and addr & set(cur):
                print(f"{name}: line {ln_no}: address s{sorted(addr & set(cur))} is the destination of a load in "
                      f"flight: {s}")
                bad += 1
            # (several loads into one register may be in flight at once: the line warm-up does that on purpose)
            for r in dst:
                cur.setdefault(r, ln_no)
            continue
        if cur:
            clash = all_sgprs(s) & set(cur)
            if clash and report:
                print(f"{name}: line {ln_no}: touches s{sorted(clash)} (loaded at line "
                      f"{min(cur[r] for r in clash)}) before the wait: {s}")
                bad += 1
    return cur, bad


def check(path: str) -> int:
    txt = open(path).read()
    parts = re.split(r"\n(_ZN4fsmc\d+decode_kernel\w+):[^\n]*\n", txt)
    bad = 0
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
        blocks, by_name = blocks_of(body)
        state_in = {b.name: {} for b in blocks}
        work = [blocks[0].name]
        seen = set()
        while work:
            bn = work.pop()
            b = by_name[bn]
            out, _ = transfer(b, state_in[bn], False, name)
            for sn in b.succ:
                if sn not in by_name:
                    continue
                merged = dict(state_in[sn])
                changed = sn not in seen
                for r, l in out.items():
                    if r not in merged:
                        merged[r] = l
                        changed = True
                if changed:
                    state_in[sn] = merged
                    seen.add(sn)
                    work.append(sn)
        for b in blocks:
            _, n = transfer(b, state_in[b.name], True, name)
            bad += n
    return bad


if __name__ == "__main__":
    n = check(sys.argv[1] if len(sys.argv) > 1 else "/tmp/fsmc_isa.s")
    print(f"{n} violation(s)")
    sys.exit(1 if n else 0)
