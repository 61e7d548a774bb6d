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
    "-fno-slp-vectorize",  # packed-f32 SLP of the k-recurrences costs more moves than it saves
    "-fPIC", "-shared",
    "-Wno-pass-failed",  # the generic (runtime-K) instantiation cannot unroll its k-loops, by design
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


HOST_DIR = os.path.join(CSRC, "host")
HOST_SOURCES = ["decoding_quantities.cpp", "decoding_params.cpp", "data.cpp", "hmm.cpp", "hashing.cpp", "drivers.cpp",
                "pybind_module.cpp"]


def host_module_path() -> str:
    import sysconfig

    return os.path.join(HERE, "_pyasmc" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_host(force: bool = False) -> str:
    """g++ builds the host orchestration + pybind11 module; it links the HIP library through its C ABI."""
    import sysconfig

    import pybind11

    out = host_module_path()
    srcs = [os.path.join(HOST_DIR, f) for f in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST_DIR, f) for f in os.listdir(HOST_DIR) if f.endswith(".hpp")]
    deps.append(os.path.join(ROOT, "include", "fastsmc_hip.h"))
    if not force and _newer(out, deps) and os.path.getmtime(out) >= os.path.getmtime(HIP_LIB):
        return out
    objs = []
    inc = ["-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"]]
    flags = ["-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall"]
    procs = []
    for s in srcs:
        o = os.path.join(HOST_DIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not _newer(o, deps):
            procs.append(subprocess.Popen(["g++", *flags, *inc, "-c", s, "-o", o], cwd=ROOT))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("host compilation failed")
    subprocess.run(["g++", "-shared", "-o", out, *objs, "-L", HERE, "-lfastsmc_hip", "-lz",
                    "-Wl,-rpath,$ORIGIN"], check=True, cwd=ROOT)
    return out
