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


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_no_instruction_touches_a_scalar_load_in_flight(tmp_path):
    sys.path.insert(0, ROOT)
    from fastsmc_amd.build import HIPCC_FLAGS

    out = str(tmp_path / "fsmc.s")
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run(["hipcc", *flags, "-S", "--cuda-device-only", "-Wno-unused-command-line-argument", "-o", out,
                    os.path.join(ROOT, "fastsmc_amd", "csrc", "fsmc_capi.hip")], check=True)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_inflight_sgprs as chk

    assert chk.check(out) == 0
