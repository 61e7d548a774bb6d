"""One HIP runtime per process: with the library (linked against the system ROCm) loaded BEFORE a PyTorch wheel that
bundles its own libamdhip64, two runtimes end up mapped and the one-wave-per-SIMD kernels fail to launch with an opaque
error.  fsmc_ctx_create tries the query that breaks and, when it does, looks at the mapped files and says what
happened (FSMC_ERUNTIME: a C-ABI consumer gets a diagnosis instead of "unknown error"); a process that maps two copies
and can still launch is not refused.  Run in a child process: the parent keeps its single runtime."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, os, sys
lib = C.CDLL(os.path.join(sys.argv[1], "fastsmc_amd", "libfastsmc_hip.so"))   # system ROCm runtime comes with it
import torch                                                                   # ... and torch brings its own
paths = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64.so" in l}
lib.fsmc_last_error.restype = C.c_char_p
h = C.c_void_p()
rc = lib.fsmc_ctx_create(0, None, C.byref(h))
print(len(paths), rc, (lib.fsmc_last_error(None) or b"").decode())
"""


def test_two_runtimes_that_break_the_launch_get_a_diagnosis():
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    n_paths, rc, msg = r.stdout.strip().split(" ", 2)
    if int(n_paths) < 2:
        pytest.skip("this image resolves both to one libamdhip64: nothing to refuse")
    if int(rc) == 0:
        return  # two copies mapped and the kernels launch all the same: a diagnosis is not a gate
    assert int(rc) == -8 and "two HIP runtimes" in msg  # FSMC_ERUNTIME
