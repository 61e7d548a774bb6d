"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/fastsmc_hip.h declares, and fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import pytest

from fastsmc_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fastsmc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fsmc_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(capi.SYMBOLS)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g

    g.build()
    lib = capi.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.FsmcError) as ei:
        capi.Context(0)
    assert ei.value.code == -2  # FSMC_ENODEVICE
    assert "no CPU fallback" in str(ei.value)
