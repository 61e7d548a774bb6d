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
    "-Wno-pass-failed",  # (loops the unroller gives up on: segment ages walk memory with real loops, by design)
]

# family members of the library (csrc/fsmc_instances.h): one translation unit each, compiled in parallel
KT_MEMBERS = [16, 32, 48, 64, 69, 80, 96, 112, 128]
# exact (ghost-free) members beside the 69-state one (csrc/fsmc_instances.h, FSMC_EXACT_KT): a model of exactly that many
# states pays for no padding.  FSMC_EXACT_MEMBERS="50 100 75" in the environment of a build lists others (none a multiple
# of 16, each <= 128 states); the list is part of the library's source hash.
EXACT_MEMBERS = sorted(set(int(x) for x in os.environ.get("FSMC_EXACT_MEMBERS", "50 100").split()))  # (a member once)
# wave-group kernel: (states per wave, waves per group) -- csrc/fsmc_instances.h, FSMC_ALL_W2
W2_MEMBERS = [(48, 4), (64, 4), (80, 4), (64, 6), (64, 7), (64, 8), (80, 8), (96, 8), (128, 8)]


def w2_unit_name(kh: int, nw: int) -> str:
    return f"w2_{kh}" if nw == 4 else f"w2_{kh}x{nw}"


def w2_units() -> list[tuple[str, list[str]]]:
    """(unit name, -D switches) of the wave-group kernel's translation units: one per member, two (array mode, sequence
    mode) for the members beyond 512 states, whose ten kernels take minutes to compile."""
    out = []
    for kh, nw in W2_MEMBERS:
        defs = [f"-DFSMC_INSTANCE_W2={kh}", f"-DFSMC_INSTANCE_NW={nw}"]
        if kh * nw > 512:
            out.append((w2_unit_name(kh, nw), defs + ["-DFSMC_INSTANCE_SEQ=0"]))
            out.append((w2_unit_name(kh, nw) + "_seq", defs + ["-DFSMC_INSTANCE_SEQ=1"]))
        else:
            out.append((w2_unit_name(kh, nw), defs))
    return out
OBJ_DIR = os.path.join(CSRC, "obj")


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def hip_sources() -> list[str]:
    """Everything the HIP library is built from: every .hip / .h under csrc/ (not host/) and the public header."""
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    srcs.append(os.path.join(ROOT, "include", "fastsmc_hip.h"))
    return srcs


def hip_source_hash() -> str:
    """Hash of the HIP library's sources -- every .hip / .h under csrc/, the public header -- and of the compiler flags:
    profiles/*_traffic.json carry it so that a measurement of another build is not reported for this one (bench.py)."""
    import hashlib

    h = hashlib.sha256()
    h.update(" ".join(HIPCC_FLAGS + exact_define()).encode())  # (a build with other flags or members is another library)
    for s in hip_sources():
        h.update(os.path.basename(s).encode())
        h.update(open(s, "rb").read())
    return h.hexdigest()[:16]


def exact_define() -> list[str]:
    """-D for a list of exact members other than the header's default."""
    if EXACT_MEMBERS == [50, 100]:
        return []
    bad = [k for k in EXACT_MEMBERS if k % 16 == 0 or not 0 < k <= 128 or k == 69]
    if bad:
        raise ValueError(f"FSMC_EXACT_MEMBERS: {bad} -- an exact member has 1..128 states, not a multiple of 16 (69 is built in)")
    return ["-DFSMC_EXACT_KT(Y)=" + " ".join(f"Y({k})" for k in EXACT_MEMBERS)]


def build_hip(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    """hipcc cross-compiles the gfx950 code objects without a GPU: fsmc_capi.hip (host side + kernel selection),
    fsmc_identify_sort.hip and fsmc_identify_seeds.hip (rocPRIM sorts of the identification step) and fsmc_inst.hip once per family member, in parallel, linked into one shared library."""
    srcs = hip_sources()
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, "members.txt")  # (another list of exact members is another library)
    members = " ".join(str(k) for k in KT_MEMBERS + EXACT_MEMBERS + [n for n, _ in w2_units()])
    if not force and _newer(HIP_LIB, srcs) and os.path.exists(stamp) and open(stamp).read() == members:
        return HIP_LIB
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"] + exact_define() + ["-c"]
    if verbose:
        cflags.append("-Rpass-analysis=kernel-resource-usage")
    units = [("capi", os.path.join(CSRC, "fsmc_capi.hip"), []),
             ("idsort", os.path.join(CSRC, "fsmc_identify_sort.hip"), []),
             ("idseeds", os.path.join(CSRC, "fsmc_identify_seeds.hip"), [])]
    units += [(f"kt{k}", os.path.join(CSRC, "fsmc_inst.hip"), [f"-DFSMC_INSTANCE_KT={k}"])
              for k in KT_MEMBERS + EXACT_MEMBERS]
    units += [(name, os.path.join(CSRC, "fsmc_inst.hip"), defs) for name, defs in w2_units()]
    # longest first (the wide members take a minute or more each, the small ones seconds): the queue's tail is short
    def cost(u):
        if u[0].startswith("w2_"):
            kh, _, nw = u[0][3:].replace("_seq", "").partition("x")
            return 1000 + int(kh) * int(nw or 4) + (1 if u[0].endswith("_seq") else 0)
        return int(u[0][2:]) if u[0][2:].isdigit() else 60
    units.sort(key=cost, reverse=True)
    jobs = jobs or max(1, min(len(units), os.cpu_count() or 1))
    pending = list(units)
    running: list[tuple[str, subprocess.Popen]] = []
    objs = []
    failed = []
    while pending or running:
        while pending and len(running) < jobs:
            name, src, defs = pending.pop(0)
            obj = os.path.join(OBJ_DIR, name + ".o")
            objs.append(obj)
            running.append((name, subprocess.Popen(["hipcc", *cflags, *defs, "-o", obj, src], cwd=ROOT)))
        name, proc = running.pop(0)
        if proc.wait() != 0:
            failed.append(name)
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB, *objs], check=True, cwd=ROOT)
    for f in os.listdir(OBJ_DIR):  # objects of units the library no longer has (tools/build_variant.py links what is here)
        if f.endswith(".o") and os.path.join(OBJ_DIR, f) not in objs:
            os.remove(os.path.join(OBJ_DIR, f))
    open(stamp, "w").write(members)
    return HIP_LIB


HOST_DIR = os.path.join(CSRC, "host")
HOST_SOURCES = ["decoding_quantities.cpp", "decoding_params.cpp", "data.cpp", "hmm.cpp", "hashing.cpp", "drivers.cpp",
                "pybind_module.cpp", "pybind_containers.cpp"]


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
