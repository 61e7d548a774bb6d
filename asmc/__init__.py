"""``asmc`` -- the reference's Python package name over the MI355X engine.

A script written against PalamaraLab/FastSMC's ``asmc`` package (``ASMC_SRC/SRC/__init__.py:18-33`` re-exports
the pybind11 module ``pyASMC``) imports the same names from here; they are the classes of
``fastsmc_amd._pyasmc`` (host orchestration in C++, decode on the GPU through ``libfastsmc_hip.so``).
"""
from fastsmc_amd.api import (  # noqa: F401
    ASMC,
    BinaryDataReader,
    Data,
    DecodePairsReturnStruct,
    DecodingMode,
    DecodingModeOverall,
    DecodingParams,
    DecodingQuantities,
    DecodingReturnValues,
    FastSMC,
    HMM,
    IbdPairDataLine,
    Individual,
    PairObservations,
)

__all__ = [
    "BinaryDataReader", "DecodingModeOverall", "DecodingMode", "DecodingReturnValues", "DecodePairsReturnStruct",
    "IbdPairDataLine", "Individual", "PairObservations", "DecodingQuantities", "DecodingParams", "Data", "HMM",
    "FastSMC", "ASMC",
]
