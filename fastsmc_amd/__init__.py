"""fastsmc_amd -- MI355X-native pairwise coalescent-HMM decode (the hot path of PalamaraLab/FastSMC).

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI of ``include/fastsmc_hip.h``),
``capi`` (ctypes binding of that ABI), ``synth`` (synthetic inputs) and the host-side mirror of the
reference interface.  See DESIGN.md.
"""
__version__ = "0.1.0"


def _one_hip_runtime_per_process() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  If this package's libraries (linked
    against the system ROCm) are loaded first and torch is imported afterwards, BOTH runtimes end up in the process;
    the kernels that are built for one wave per SIMD (K > 192) then fail to launch ("unknown error" from the
    occupancy query; measured on this image: torch 2.10+rocm7.0 next to ROCm 7.2).  Importing torch first makes the
    dynamic loader resolve our DT_NEEDED libamdhip64.so.7 to the copy that is already loaded -- one runtime.  Without
    torch installed there is nothing to do."""
    import importlib.util
    import sys

    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        try:
            import torch  # noqa: F401
        except Exception:  # a broken torch must not take this package down with it
            pass


_one_hip_runtime_per_process()
