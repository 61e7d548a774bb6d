"""Python-facing helpers over the pybind11 module ``_pyasmc`` (the reference's ``asmc`` package surface).

``from fastsmc_amd.api import *`` gives the reference names: ``DecodingParams``, ``DecodingQuantities``,
``Data``, ``HMM``, ``ASMC``, ``FastSMC``, ``DecodingModeOverall``, ``DecodingMode`` ...
"""
from __future__ import annotations

import numpy as np

from . import _pyasmc
from ._pyasmc import (ASMC, BinaryDataReader, Data, DecodePairsReturnStruct, DecodingMode, DecodingModeOverall,  # noqa: F401
                      DecodingParams, DecodingQuantities, DecodingReturnValues, FastSMC, HMM, IbdPairDataLine, Individual,
                      Match, PairObservations, cmBetween, hashingCandidates, hashingCandidatesDevice, hashingWords)

__all__ = ["ASMC", "BinaryDataReader", "IbdPairDataLine", "Data", "DecodePairsReturnStruct", "DecodingMode", "DecodingModeOverall", "DecodingParams",
           "DecodingQuantities", "DecodingReturnValues", "FastSMC", "HMM", "Individual", "PairObservations", "Match", "cmBetween",
           "hashingCandidates", "hashingCandidatesDevice", "hashingWords",
           "decoding_quantities_from_tables", "PreparedModelView"]


def decoding_quantities_from_tables(t) -> DecodingQuantities:
    """Wrap a ``fastsmc_amd.synth.ModelTables`` as a ``DecodingQuantities`` without going through a file."""
    return DecodingQuantities.from_arrays(
        int(t.csfs_samples), t.discretization, t.expected_times, t.initial_state_prob, t.column_ratios, t.keys,
        t.D, t.B, t.U, t.RR, t.compressed_emission, t.classic_emission, t.folded_ascertained_csfs, t.ascertained_csfs,
        t.csfs, t.folded_csfs, t.homozygous_keys if t.homozygous_keys.size else None,
        t.homozygous if t.homozygous_keys.size else None)


class PreparedModelView:
    """Attribute view of ``HMM.preparedModel()`` (the argument ``capi.Context.create_model`` expects)."""

    def __init__(self, d: dict):
        self.__dict__.update(d)
        self.probability_threshold = np.float32(d["probability_threshold"])
