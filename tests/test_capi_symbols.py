"""CPU: the C-ABI library loads and exports every symbol include/psgpu.h declares;
without a GPU every compute entry point fails loudly (no CPU fallback)."""
import os
import re

import pytest

import pocketsphinx_amd as P
from pocketsphinx_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "psgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psgpu_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    P.build_library()
    L = P.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libpsgpu.so does not export %s" % n
    assert sorted(capi.SYMBOLS) == names


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import pso
    with pytest.raises(P.PsgpuError):
        P.PtmModel(pso.load_tables())
