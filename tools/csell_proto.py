"""Prototype driver for tools/csell_proto.hip: builds the compact sliced-ELL operand of the sweep matrix (100k x 100k, 1 %)
with torch on the GPU, runs the lane-per-row kernel at B = 16 / 32 / 64, checks it against the library's ss_spmm_f32 and
prints both times.   python tools/csell_proto.py [B ...]      env: N (100000), DENS (0.01), DEPTH (4), CGS ("auto")"""
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import torch

import simspread_jl_amd as ss
from simspread_jl_amd import _lib as L


def build_lib():
    so = os.path.join(HERE, "csell_proto.so")
    src = os.path.join(HERE, "csell_proto.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-Wno-unused-value", "-o", so, src])
    return C.CDLL(so)


def rand_csr(M, K, dens, gen, dev):
    n_draw = int(M * K * dens)
    rows = torch.randint(0, M, (n_draw,), device=dev, generator=gen, dtype=torch.int64)
    cols = torch.randint(0, K, (n_draw,), device=dev, generator=gen, dtype=torch.int64)
    keys = torch.unique(rows * K + cols)
    r = torch.div(keys, K, rounding_mode="floor")
    c = keys - r * K
    return r, c


def build_csell(r, c, v, M, K, KC, QT):
    """(row, col, val) of every non-zero (any order) -> the block arrays of csell_proto.hip."""
    dev = r.device
    S = (M + 63) // 64
    nchunks = (K + KC - 1) // KC
    NPC = QT // 4
    NC = max(1, 64 // QT)                       # tile rows per 256-byte LDS line
    ch = torch.div(c, KC, rounding_mode="floor")
    kl = c - ch * KC
    lane = r % 64
    # order inside a sub-row: lanes that read the same 16-byte slot in the same cycle start in different row classes
    q = torch.div(lane, NPC, rounding_mode="floor") % NC
    okey = (kl % NC - q) % NC
    key = ((ch * M + r) * NC + okey) * KC + kl
    order = torch.argsort(key)
    del key
    r, ch, kl, v, lane = r[order], ch[order], kl[order], v[order], lane[order]
    del order
    grp = ch * M + r                              # sub-row id, ascending
    first = torch.ones_like(grp, dtype=torch.bool)
    first[1:] = grp[1:] != grp[:-1]
    idx = torch.arange(grp.numel(), device=dev)
    start = torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), 0).values
    t = idx - start                               # rank inside the sub-row
    del first, start, idx
    u = torch.div(t, 2, rounding_mode="floor")
    e = t - 2 * u
    sl = torch.div(r, 64, rounding_mode="floor")
    blk = ch * S + sl
    umax = int(u.max().item()) + 1
    assert umax <= 255, umax
    key2 = (((blk * umax + u) * 64 + lane) * 2 + e)
    order = torch.argsort(key2)
    del key2
    blk, u, lane, e, kl, v = blk[order], u[order], lane[order], e[order], kl[order], v[order]
    del order
    pairpos = torch.cumsum((e == 0).to(torch.int64), 0) - 1
    npairs = int(pairpos[-1].item()) + 1
    pidx_lo = torch.full((npairs + 64,), KC, dtype=torch.int64, device=dev)
    pidx_hi = torch.full((npairs + 64,), KC, dtype=torch.int64, device=dev)
    pval = torch.zeros((npairs + 64, 2), dtype=torch.float32, device=dev)
    m0 = e == 0
    pidx_lo[pairpos[m0]] = kl[m0]
    pval[pairpos[m0], 0] = v[m0]
    m1 = ~m0
    pidx_hi[pairpos[m1]] = kl[m1]
    pval[pairpos[m1], 1] = v[m1]
    pidx = (pidx_lo | (pidx_hi << 16)).to(torch.int32)      # KC < 2^15: no wrap
    nblk = nchunks * S
    pairs_per_blk = torch.bincount(blk[m0], minlength=nblk)
    blk_base = torch.zeros(nblk, dtype=torch.int64, device=dev)
    blk_base[1:] = torch.cumsum(pairs_per_blk, 0)[:-1]
    np_lane = torch.zeros(nblk * 64, dtype=torch.int64, device=dev)
    np_lane.scatter_reduce_(0, (blk * 64 + lane)[m0], (u + 1)[m0], "amax", include_self=True)
    blk_max = np_lane.view(nblk, 64).max(1).values
    desc = torch.zeros((nblk + 1, 2), dtype=torch.int32, device=dev)      # + one empty block (past the end of every wave's list)
    desc[:nblk, 0] = blk_base.to(torch.int32)
    desc[:nblk, 1] = blk_max.to(torch.int32)
    desc[nblk, 0] = npairs
    np2 = torch.zeros((nblk + 1) * 64, dtype=torch.uint8, device=dev)
    np2[:nblk * 64] = np_lane.to(torch.uint8)
    return dict(S=S, nchunks=nchunks, blk_base=blk_base.to(torch.int32), blk_np=np_lane.to(torch.uint8),
                blk_max=blk_max.to(torch.uint8), pidx=pidx.contiguous(), pval=pval.contiguous(), npairs=npairs,
                desc=desc.contiguous(), np2=np2)


def choose_cut(S, nchunks, max_spw, ncu=256):
    best = None
    for spw in range(1, max_spw + 1):
        rb = (S + spw - 1) // spw
        for cg in range(1, min(nchunks, 16) + 1):
            g = rb * cg
            rounds = (g + ncu - 1) // ncu
            work = rounds * spw * ((nchunks + cg - 1) // cg)      # block steps on the busiest CU
            stage = rounds * ((nchunks + cg - 1) // cg) * 40       # restaging, in the same unit (rough)
            cost = work + stage + 2 * cg
            if best is None or cost < best[0]:
                best = (cost, spw, rb, cg)
    return best[1], best[2], best[3]


def main():
    widths = [int(x) for x in sys.argv[1:]] or [16, 32, 64]
    M = K = int(os.environ.get("N", 100_000))
    dens = float(os.environ.get("DENS", 0.01))
    depth = int(os.environ.get("DEPTH", 4))
    binary = os.environ.get("BINARY", "0") == "1"
    lib = build_lib()
    ss.init(0)
    ss.use_torch_stream()
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev)
    gen.manual_seed(20250222 + 3)
    r, c = rand_csr(M, K, dens, gen, dev)
    nnz = r.numel()
    v = torch.rand(nnz, device=dev, dtype=torch.float32, generator=gen) + 0.5
    if binary:
        v.fill_(1.0)
    ptr = torch.zeros(M + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(torch.bincount(r, minlength=M), 0)
    h = C.c_void_p()
    slib = L.lib()
    L.check(slib.ss_spmat_create_csr_f32(M, K, ptr.data_ptr(), c.to(torch.int32).data_ptr(), v.data_ptr(), 0, L.SS_MEM_DEVICE, C.byref(h)))
    for B in widths:
        QT = B
        NPS = {16: 4, 32: 2, 64: 1}[QT]
        KC = int(os.environ.get("KC", (160 * 1024) // (QT * 4) - 1))
        fmt = build_csell(r, c, v, M, K, KC, QT)
        S, nchunks = fmt["S"], fmt["nchunks"]
        spw, rbn, cg = choose_cut(S, nchunks, 16 * NPS)
        if os.environ.get("CUT"):
            spw, cg = (int(x) for x in os.environ["CUT"].split(","))
            rbn = (S + spw - 1) // spw
        R = torch.rand(K, B, device=dev, dtype=torch.float32, generator=gen)
        F = torch.zeros(M, B, device=dev, dtype=torch.float32)
        P = torch.empty(cg * M * QT if cg > 1 else 4, device=dev, dtype=torch.float32)
        ms = C.c_float(0)
        torch.cuda.synchronize()
        rc = lib.csell_run(C.c_void_p(fmt["blk_base"].data_ptr()), C.c_void_p(fmt["blk_np"].data_ptr()),
                           C.c_void_p(fmt["blk_max"].data_ptr()), C.c_void_p(fmt["pidx"].data_ptr()),
                           C.c_void_p(fmt["pval"].data_ptr()), C.c_int64(M), C.c_int64(K), KC, nchunks, S,
                           C.c_void_p(R.data_ptr()), C.c_int64(B), C.c_void_p(F.data_ptr()), C.c_int64(B),
                           C.c_void_p(P.data_ptr()), spw, rbn, cg, QT, 1 if binary else 0, depth, 10, C.byref(ms))
        assert rc == 0, rc
        F1 = F.clone()
        F.zero_()
        ms2 = C.c_float(0)
        rc = lib.csell2_run(C.c_void_p(fmt["desc"].data_ptr()), C.c_void_p(fmt["np2"].data_ptr()), C.c_void_p(fmt["pidx"].data_ptr()),
                            C.c_void_p(fmt["pval"].data_ptr()), C.c_int64(M), C.c_int64(K), KC, nchunks, S,
                            C.c_void_p(R.data_ptr()), C.c_int64(B), C.c_void_p(F.data_ptr()), C.c_int64(B),
                            C.c_void_p(P.data_ptr()), spw, rbn, cg, QT, 1 if binary else 0, depth, 10, C.byref(ms2))
        assert rc == 0, rc
        v2_equal = bool(torch.equal(F, F1))
        F2 = torch.empty(M, B, device=dev, dtype=torch.float32)
        t_lib = []
        for it in range(7):
            L.check(slib.ss_spmm_f32(h, R.data_ptr(), B, B, 0, F2.data_ptr(), B, 0, L.SS_MEM_DEVICE))
            if it >= 2:
                tl = ss.timing_last()
                t_lib.append(tl["spmm_ms"] + tl["epilogue_ms"])
        torch.cuda.synchronize()
        err = float(((F - F2).abs().max() / F2.abs().max()).item())
        by = nnz * 8 + (M + 1) * 4 + K * B * 4 + M * B * 4
        stream = fmt["npairs"] * 12 + fmt["blk_np"].numel()
        print(json.dumps({"B": B, "abl": os.environ.get("CSELL_ABL", "0"), "KC": KC, "chunks": nchunks, "cut": {"slices_per_wg": spw, "RB": rbn, "CG": cg},
                          "csell_ms": round(ms.value, 4), "csell2_ms": round(ms2.value, 4), "v2_bitwise_equal_v1": v2_equal, "lib_ms": round(float(np.mean(t_lib)), 4), "lib_path": ss.path_last(),
                          "frac_hbm_csell": round(by / (ms.value * 1e-3) / 8e12, 4), "max_rel_diff": err,
                          "pairs": fmt["npairs"], "pad_entries_frac": round(2 * fmt["npairs"] / nnz - 1, 4),
                          "stream_bytes_per_nnz": round(stream / nnz, 2),
                          "steps_executed_over_needed": round(float(fmt["blk_max"].sum().item()) * 128 / nnz, 3)}), flush=True)
        del fmt, R, F, F2, P
    slib.ss_spmat_destroy(h)


if __name__ == "__main__":
    main()
