"""ctypes binding of libsimspread_hip.so (C ABI in include/simspread_hip.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device can be
initialised, every compute entry point raises.  Loading the library and listing its
symbols works without a GPU (used by the CPU-side tests).
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SS_LIB_PATH") or os.path.join(_HERE, "libsimspread_hip.so")   # SS_LIB_PATH: A/B builds of the same ABI
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "simspread_hip.h")

SS_OK = 0
SS_MEM_HOST, SS_MEM_DEVICE = 0, 1
SS_ROWS_QUERY, SS_ROWS_SOURCE = 0, 1
SS_LAYOUT_ROWMAJOR, SS_LAYOUT_COLMAJOR = 0, 1

_ERR_NAMES = {-1: "SS_EINVAL", -2: "SS_ENOMEM", -3: "SS_EHIP", -4: "SS_ENODEV", -5: "SS_EUNSUPPORTED"}


class SimSpreadError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {message}")
        self.code = code


_lib = None
_inited_device = None

_i64, _i32, _int = C.c_int64, C.c_int32, C.c_int
_vp = C.c_void_p


def _sigs():
    f32, f64 = C.c_float, C.c_double
    s = {
        "ss_version": ([], _int),
        "ss_source_hash": ([], C.c_char_p),
        "ss_last_error": ([], C.c_char_p),
        "ss_device_count": ([], _int),
        "ss_init": ([_int], _int),
        "ss_shutdown": ([], _int),
        "ss_set_stream": ([_vp], _int),
        "ss_reset_stream": ([], _int),
        "ss_synchronize": ([], _int),
        "ss_timing_last": ([_vp, _int], _int),
        "ss_timing_hold": ([_int], _int),
        "ss_path_last": ([_vp, _int], _int),
        "ss_comm_unique_id": ([_vp], _int),
        "ss_comm_init": ([_vp, _int, _int], _int),
        "ss_comm_destroy": ([], _int),
        "ss_comm_info": ([_vp, _vp], _int),
        "ss_gather_rows_f32": ([_vp, _i64, _vp, _vp, _int], _int),
        "ss_gather_rows_f64": ([_vp, _i64, _vp, _vp, _int], _int),
        "ss_graph_destroy": ([_vp], _int),
        "ss_graph_info": ([_vp, _vp], _int),
        "ss_graph_degrees": ([_vp, _vp, _vp, _vp], _int),
        "ss_spmat_destroy": ([_vp], _int),
        "ss_spmat_cost": ([_vp, _i64, _vp, _vp], _int),
    }
    s["ss_topl_f32"] = ([_vp, _i64, _i64, _i64, _int, _vp, _vp, _int], _int)
    s["ss_rank_metrics_f32"] = ([_vp, _vp, _i64, f64, _vp, _int], _int)
    s["ss_graph_create_similarity_f32"] = ([_i64] * 3 + [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _int, f32, _int, _int, _vp], _int)
    s["ss_graph_create_similarity_f64"] = ([_i64] * 3 + [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _int, f64, _int, _int, _vp], _int)
    for suf, ft in (("f32", f32), ("f64", f64)):
        s[f"ss_cutoff_{suf}"] = ([_vp, _i64, _i64, _i64, ft, _int, _vp, _i64, _int], _int)
        s[f"ss_similarity_jaccard_{suf}"] = ([_vp, _i64, _i64, _i64, _vp, _i64, _int], _int)
        s[f"ss_row_degree_{suf}"] = ([_vp, _i64, _i64, _i64, _vp, _int], _int)
        s[f"ss_spread_{suf}"] = ([_vp, _i64, _i64, _i64, _vp, _i64, _int], _int)
        s[f"ss_graph_create_csr_{suf}"] = ([_i64] * 4 + [_vp] * 9 + [_int, _int, _vp], _int)
        s[f"ss_graph_create_dense_{suf}"] = ([_i64] * 4 + [_vp, _i64, _vp, _i64, _vp, _i64, _int, ft, _int, _int, _vp], _int)
        s[f"ss_graph_create_general_{suf}"] = ([_i64] * 3 + [_vp] * 9 + [_int, _int, _vp], _int)
        s[f"ss_predict_{suf}"] = ([_vp, _int, _i64, _i64, _int, _vp, _i64, _int, _int], _int)
        s[f"ss_predict_loo_{suf}"] = ([_vp, _i64, _i64, _int, _vp, _i64, _int, _int], _int)
        s[f"ss_predict_kfold_{suf}"] = ([_vp, _vp, _int, _int, _vp, _i64, _int, _int], _int)
        s[f"ss_spmat_create_csr_{suf}"] = ([_i64, _i64, _vp, _vp, _vp, _int, _int, _vp], _int)
        s[f"ss_spmm_{suf}"] = ([_vp, _vp, _i64, _i64, _int, _vp, _i64, _int, _int], _int)
    return s


SIGNATURES = _sigs()


def header_symbols():
    """Every function the C header declares (used by the symbol-export test)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z0-9_]+)\s*\(", text)))


def load():
    """dlopen the library (no GPU needed for this step)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C simspread.jl_amd/csrc).  There is no CPU fallback for the HIP path.")
    lib = C.CDLL(LIB_PATH)
    for name, (args, res) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    # a library older than the sources next to it must not run silently (its results and profiles would be credited to
    # kernels that did not produce them)
    if os.path.isdir(os.path.join(_HERE, "csrc")) and os.environ.get("SS_ALLOW_STALE_LIB") != "1":
        built = lib.ss_source_hash().decode()
        have = source_hash()
        # (a library whose build could not compute the hash -- no python3 next to make -- says "" or "unknown": accepted)
        if re.fullmatch(r"[0-9a-f]{12}", built) and built != have:
            raise ImportError(f"{LIB_PATH} was built from other kernel sources ({built}) than the ones in csrc/ ({have}): "
                              "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def check(rc: int):
    if rc != SS_OK:
        raise SimSpreadError(rc, load().ss_last_error().decode("utf-8", "replace"))


def init(device: int | None = None):
    """Bind this process to one GPU (idempotent).  Raises if no gfx950 device is usable."""
    global _inited_device
    lib = load()
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if _inited_device is None else _inited_device
    if _inited_device == device:
        return lib
    check(lib.ss_init(int(device)))
    _inited_device = device
    return lib


def lib():
    """The initialised library; the compute paths call this, so they fail loudly without a GPU."""
    return init()


def use_torch_stream():
    """Enqueue on torch's current stream: required whenever torch CUDA tensors are handed over by
    pointer, so that the kernels that produced them are ordered before the library's."""
    import torch
    check(lib().ss_set_stream(torch.cuda.current_stream().cuda_stream))


def timing_hold(enable: bool = True) -> None:
    """Let the stage timings of successive calls add up (read the sums with timing_last) instead of replacing one
    another, so that a loop of predictions need not stop after every call."""
    check(lib().ss_timing_hold(1 if enable else 0))


def timing_last():
    import numpy as np
    buf = np.zeros(8, dtype=np.float64)
    check(lib().ss_timing_last(buf.ctypes.data, 8))
    return dict(total_ms=buf[0], transfer_ms=buf[1], spmm_ms=buf[2], epilogue_ms=buf[3], h2d_ms=buf[4],
                d2h_ms=buf[5], spmm_launches=int(buf[6]), transfer_launches=int(buf[7]))


def path_last() -> list:
    """Kernel tags the last predict / spmm call of this thread went through (ss_path_last)."""
    buf = C.create_string_buffer(512)
    check(lib().ss_path_last(C.cast(buf, C.c_void_p), 512))
    return [t for t in buf.value.decode().split(",") if t]


def source_hash() -> str:
    """Short hash of the kernel sources (csrc/*.hip, *.hpp): profiles/ records it next to the PMC numbers so that a
    consumer (bench.py) can tell whether the counters were taken from the kernels it is running."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()[:12]
