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


def test_entry_points_are_safe_from_several_threads_without_a_device():
    """Handle-scoped state, per-thread error messages: several host threads hammering entry points that must fail
    (no device initialised here) get their own message each and nothing crashes or deadlocks.  The two-handles-in-
    two-threads run on a real device is tests/test_gpu_parity.py::test_two_threads_two_handles."""
    import ctypes as C
    import threading
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the GPU test")
    lib = _lib.load()
    errors, msgs = [], {}

    def worker(tid):
        try:
            for i in range(200):
                if tid % 2 == 0:
                    rc = lib.ss_synchronize()
                    want = "ss_init"
                else:
                    rc = lib.ss_path_last(None, 0)
                    want = "ss_path_last"
                msg = lib.ss_last_error().decode()
                if rc == 0 or want not in msg:
                    errors.append((tid, i, rc, msg))
            msgs[tid] = msg
        except Exception as e:  # pragma: no cover
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in threads), "deadlock"
    assert not errors, errors[:3]
    assert "ss_init" in msgs[0] and "ss_path_last" in msgs[1]


def test_no_device_side_abort_in_any_kernel_source():
    """include/simspread_hip.h promises that no entry point throws or aborts: a violated kernel assumption must be a
    host-side check that returns an SS_E* code (e.g. the static-LDS check in launch_spmm_sell), never a trap on the GPU."""
    import re
    root = os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc")
    bad = []
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".hpp")):
            with open(os.path.join(root, f)) as fh:
                for ln, line in enumerate(fh, 1):
                    code = line.split("//")[0]
                    if re.search(r"__builtin_trap|\babort\s*\(|\bassert\s*\(|__assert_fail|std::terminate|\bthrow\b", code):
                        bad.append(f"{f}:{ln}: {line.strip()}")
    assert not bad, bad


def test_library_reports_the_sources_it_was_built_from_and_a_stale_one_is_refused(monkeypatch):
    """ss_source_hash() (csrc/build_id.cpp, value passed by the Makefile) must equal the hash of the sources next to the
    library; the loader refuses a library built from other sources instead of running it silently."""
    from simspread_jl_amd import _lib
    lib = _lib.load()
    assert lib.ss_source_hash().decode() == _lib.source_hash()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "source_hash", lambda: "0" * 12)
    with pytest.raises(ImportError, match="other kernel sources"):
        _lib.load()
    monkeypatch.setenv("SS_ALLOW_STALE_LIB", "1")
    assert _lib.load() is not None
