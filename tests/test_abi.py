"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/simspread_hip.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import os

import pytest

import simspread_jl_amd as ss
from simspread_jl_amd import _lib


def test_library_is_built_in_tree():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    assert os.path.dirname(_lib.LIB_PATH).endswith("simspread.jl_amd")


def test_exports_every_header_symbol():
    lib = _lib.load()
    declared = _lib.header_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    # and the binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == declared


def test_version():
    assert _lib.load().ss_version() == 100


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ss.SimSpreadError):
        ss.cutoff(0.8, 0.5, False)


def test_product_never_imports_the_oracle():
    root = os.path.dirname(_lib.LIB_PATH)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                with open(os.path.join(dirpath, f)) as fh:
                    text = fh.read()
                assert "oracle" not in text.replace("no oracle", ""), f"{f} mentions the oracle"
