"""The Julia `ccall` layer (julia/SimSpreadHIP.jl) against the C header and the ctypes table of the Python mirror.

Julia is not installed here, so the binding cannot be executed; what CAN be checked statically is the part that goes
wrong silently at run time -- a ccall whose symbol does not exist, or whose return / argument tuple disagrees with the
C prototype.  Every `ss_*` function the header declares must have a ccall, every ccall must name a declared function,
and the three descriptions of each signature (C header, Julia tuple, ctypes table) must agree type class by type class.
The reference-compatible layer (julia/SimSpreadDevice.jl) is checked for the method table of src/SimSpread.jl:21-56
and the exact assertion messages (src/core.jl:149,156,222,223,231,314)."""
import ctypes as C
import os
import re

from simspread_jl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_signatures():
    with open(_lib.HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    sigs = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ss_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argl = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        sigs[name] = (_c_class(ret, ret=True), [_c_class(a) for a in argl])
    return sigs


def _c_class(decl, ret=False):
    d = decl.replace("const", " ").strip()
    if "*" in d or "[" in d:
        return "cstr" if (ret and "char" in d) else "ptr"
    base = d.split()[0] if not ret else d
    base = base.strip()
    return {"int": "i32", "int64_t": "i64", "int32_t": "i32", "float": "f32", "double": "f64", "void": "void"}[base.split()[0]]


def _julia_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")):
        return "ptr"
    return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "Float32": "f32", "Cfloat": "f32", "Float64": "f64",
            "Cdouble": "f64", "Cstring": "cstr", "Cvoid": "void"}[t]


def _ctypes_class(t):
    if t is None:
        return "void"
    return {C.c_int: "i32", C.c_int32: "i32", C.c_int64: "i64", C.c_float: "f32", C.c_double: "f64",
            C.c_void_p: "ptr", C.c_char_p: "cstr"}[t]


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def _julia_ccalls(path):
    with open(path) as f:
        text = f.read()
    calls = []
    for m in re.finditer(r"ccall\(\(:(ss_[a-z0-9_]+),\s*LIB\),\s*([A-Za-z0-9{}]+),\s*\(", text):
        name, ret = m.group(1), m.group(2)
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        tup = text[m.end():i - 1]
        calls.append((name, _julia_class(ret), [_julia_class(a) for a in _split_top(tup)]))
    return calls


def test_every_header_symbol_has_a_matching_ccall():
    header = _header_signatures()
    assert sorted(header) == _lib.header_symbols()
    calls = _julia_ccalls(os.path.join(ROOT, "julia", "SimSpreadHIP.jl"))
    bound = {c[0] for c in calls}
    assert bound == set(header), (sorted(set(header) - bound), sorted(bound - set(header)))
    for name, ret, args in calls:
        hret, hargs = header[name]
        assert ret == hret, (name, ret, hret)
        assert args == hargs, (name, args, hargs)


def test_ctypes_table_agrees_with_the_header_too():
    header = _header_signatures()
    for name, (args, res) in _lib.SIGNATURES.items():
        hret, hargs = header[name]
        assert _ctypes_class(res) == hret, name
        assert [_ctypes_class(a) for a in args] == hargs, name


def test_reference_method_table_and_messages_are_kept():
    with open(os.path.join(ROOT, "julia", "SimSpreadDevice.jl")) as f:
        text = f.read()
    # the hot-path part of the export list of src/SimSpread.jl:21-56
    for fn in ("k", "cutoff", "featurize", "construct", "spread", "predict", "clean!"):
        assert re.search(r"^export .*(?<![\w!])%s(?![\w!])" % re.escape(fn), text, flags=re.M), fn
    # four construct methods, three predict call forms (tuple, 3-argument, 2-argument), for Networks and NamedMatrices
    assert len(re.findall(r"^(?:function )?construct\(", text, flags=re.M)) == 4
    assert len(re.findall(r"^(?:function )?predict\(", text, flags=re.M)) >= 6
    assert "_node_groups(A::NamedMatrix, B::NamedMatrix, y::NamedMatrix)" in text
    assert "GPU::Bool=false" in text
    for msg in ("Labels and features have different number of source nodes",          # src/core.jl:149
                "Source and Features nodes have the same names!",                     # :156
                "Number of targets between test and training sets doesn't match",     # :222
                "Number of features between test and training sets doesn't match",    # :223
                "Features and drugs have the same names!",                            # :231
                "Source and feature nodes have the same names"):                      # :314
        assert msg in text, msg
    # INTEGRATION.md may only call functions that exist
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        integ = f.read()
    for fn in set(re.findall(r"\b(_[a-z_]+)\(", integ)):
        assert ("function %s(" % fn) in text or ("%s(" % fn) in text, fn
    for fn in set(re.findall(r"SimSpreadHIP\.([a-z_!]+)\(", integ + text)):
        with open(os.path.join(ROOT, "julia", "SimSpreadHIP.jl")) as f:
            low = f.read()
        assert re.search(r"^(?:function )?%s\(" % re.escape(fn), low, flags=re.M) or ("%s(" % fn) in low, fn
