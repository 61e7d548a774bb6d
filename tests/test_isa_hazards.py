"""Static check of the compiled gfx950 kernels (no GPU needed: hipcc cross-compiles): the hand-placed scalar loads
of fsmc_kernels.h are inline asm whose results arrive long after the instruction issues, which the compiler does not
know.  Between each such load and the next `s_waitcnt lgkmcnt(0)` nothing may read or write its destination
registers, and no later load may use them as its address (tools/check_inflight_sgprs.py).  A violation is a
garbage operand at best and a GPU memory fault at worst, and whether it happens depends on register allocation --
so every build is checked."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile_member(args):
    root, flags, define, out = args
    subprocess.run(["hipcc", *flags, *define.split(), "-S", "--cuda-device-only", "-Wno-unused-command-line-argument", "-o", out,
                    os.path.join(root, "fastsmc_amd", "csrc", "fsmc_inst.hip")], check=True)
    return out


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_no_instruction_touches_a_scalar_load_in_flight(tmp_path):
    sys.path.insert(0, ROOT)
    from concurrent.futures import ThreadPoolExecutor

    from fastsmc_amd.build import EXACT_MEMBERS, HIPCC_FLAGS, KT_MEMBERS, exact_define, w2_units

    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + exact_define()
    jobs = [(ROOT, flags, f"-DFSMC_INSTANCE_KT={k}", str(tmp_path / f"kt{k}.s")) for k in KT_MEMBERS + EXACT_MEMBERS]
    # (the wave-group kernel uses the same operand loads)
    jobs += [(ROOT, flags, " ".join(defs), str(tmp_path / (name + ".s"))) for name, defs in w2_units()]
    jobs.sort(key=lambda j: ("INSTANCE_W2" not in j[2], j[2]))  # (the wave-group units take longest: first)
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
        outs = list(ex.map(_compile_member, jobs))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_inflight_sgprs as chk

    for out in outs:
        assert chk.check(out) == 0, out


def test_checker_sees_a_violation_across_a_branch(tmp_path):
    """The checker itself: a register of a load in flight that is read on ONE path to the wait is a violation, the
    same read behind the wait is not, and a load whose destinations are only waited for is clean."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_inflight_sgprs as chk

    def isa(body):
        return ("\n_ZN4fsmc13decode_kernelILi1EEEvv:\n" + body + "\n.Lfunc_end0:\n")

    bad = isa("\t;;#ASMSTART\n\ts_load_dwordx2 s[4:5], s[0:1], 0x0\n\t;;#ASMEND\n"
              "\ts_cbranch_scc1 .LBB0_2\n"
              "\ts_waitcnt lgkmcnt(0)\n\ts_branch .LBB0_3\n"
              ".LBB0_2:\n\tv_mov_b32_e32 v0, s5\n\ts_waitcnt lgkmcnt(0)\n"
              ".LBB0_3:\n\tv_mov_b32_e32 v1, s4\n\ts_endpgm\n")
    good = isa("\t;;#ASMSTART\n\ts_load_dwordx2 s[4:5], s[0:1], 0x0\n\t;;#ASMEND\n"
               "\ts_cbranch_scc1 .LBB0_2\n"
               "\ts_waitcnt lgkmcnt(0)\n\ts_branch .LBB0_3\n"
               ".LBB0_2:\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v0, s5\n"
               ".LBB0_3:\n\tv_mov_b32_e32 v1, s4\n\ts_endpgm\n")
    pb, pg = tmp_path / "bad.s", tmp_path / "good.s"
    pb.write_text(bad)
    pg.write_text(good)
    assert chk.check(str(pb)) == 1
    assert chk.check(str(pg)) == 0
