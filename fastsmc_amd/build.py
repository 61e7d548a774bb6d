"""Build recipes for the native parts (used by ``__graft_entry__.build()``)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HIP_LIB = os.path.join(HERE, "libfastsmc_hip.so")

HIPCC_FLAGS = [
    "-std=c++17", "-O3", "--offload-arch=gfx950",
    "-ffp-contract=off",  # parity: the reference path has no fused multiply-add
    "-fPIC", "-shared",
]


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """hipcc cross-compiles the gfx950 code object without a GPU."""
    srcs = [os.path.join(CSRC, f) for f in ("fsmc_capi.hip", "fsmc_kernels.h")]
    srcs.append(os.path.join(ROOT, "include", "fastsmc_hip.h"))
    if not force and _newer(HIP_LIB, srcs):
        return HIP_LIB
    cmd = ["hipcc", *HIPCC_FLAGS, "-o", HIP_LIB, os.path.join(CSRC, "fsmc_capi.hip")]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True, cwd=ROOT)
    return HIP_LIB
