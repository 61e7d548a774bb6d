"""fastsmc_amd -- MI355X-native pairwise coalescent-HMM decode (the hot path of PalamaraLab/FastSMC).

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI of ``include/fastsmc_hip.h``),
``capi`` (ctypes binding of that ABI), ``synth`` (synthetic inputs) and the host-side mirror of the
reference interface.  See DESIGN.md.
"""
__version__ = "0.1.0"
