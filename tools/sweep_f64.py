"""fp64 W*R sweep at 100k x 100k, 1 % (the reference's default precision): per-B time of ss_spmm_f64 and the kernel used."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import simspread_jl_amd as ss
from simspread_jl_amd import _lib as L
ss.init(0); ss.use_torch_stream()
M = K = 100_000
g = torch.Generator(device="cuda"); g.manual_seed(20250222 + 3)
n = int(M * K * 0.01)
keys = torch.unique(torch.randint(0, M, (n,), device="cuda", generator=g) * K + torch.randint(0, K, (n,), device="cuda", generator=g))
r = torch.div(keys, K, rounding_mode="floor"); idx = (keys - r * K).to(torch.int32)
ptr = torch.zeros(M + 1, dtype=torch.int64, device="cuda"); ptr[1:] = torch.cumsum(torch.bincount(r, minlength=M), 0)
val = torch.rand(idx.numel(), device="cuda", dtype=torch.float64, generator=g) + 0.5
h = C.c_void_p()
L.check(L.lib().ss_spmat_create_csr_f64(M, K, ptr.data_ptr(), idx.data_ptr(), val.data_ptr(), 0, L.SS_MEM_DEVICE, C.byref(h)))
for B in [int(x) for x in os.environ.get("SWEEP_B", "1,4,8,16,32").split(",")]:
    R = torch.rand(K, B, device="cuda", dtype=torch.float64, generator=g); F = torch.empty(M, B, device="cuda", dtype=torch.float64)
    ms = []
    for it in range(5):
        L.check(L.lib().ss_spmm_f64(h, R.data_ptr(), B, B, 0, F.data_ptr(), B, 0, L.SS_MEM_DEVICE))
        if it >= 2:
            t = ss.timing_last(); ms.append(t["spmm_ms"] + t["epilogue_ms"])
    by = idx.numel() * 12 + (M + 1) * 4 + (K + M) * B * 8
    print(json.dumps({"B": B, "ms": round(float(np.mean(ms)), 4), "frac_hbm": round(by / (np.mean(ms) * 1e-3) / 8e12, 4), "kernel": ss.path_last()}))
