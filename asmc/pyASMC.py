"""``asmc.pyASMC`` -- the reference's extension-module name (``pybind.cpp:54``), served by ``fastsmc_amd._pyasmc``."""
from fastsmc_amd._pyasmc import *  # noqa: F401,F403
from fastsmc_amd._pyasmc import (ASMC, BinaryDataReader, Data, DecodePairsReturnStruct, DecodingMode,  # noqa: F401
                                 DecodingModeOverall, DecodingParams, DecodingQuantities, DecodingReturnValues,
                                 FastSMC, HMM, IbdPairDataLine, Individual, PairObservations)
