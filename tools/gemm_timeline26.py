"""dev: per-sub-step timeline of the 4-wave AGPR GEMM (LRP_GEMM_TILE=26), workgroup 0"""
import os, sys, torch
os.environ["LRP_GEMM_TILE"] = "26"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lxt_amd.ops as ops
from lxt_amd._lib import lib
import numpy as np
for (M, N, K, hot) in [(8192, 4096, 4096, 1), (8192, 4096, 4096, 0)]:
    if hot:
        a = torch.randn(1, K, device="cuda").bfloat16().expand(M, K); b = torch.randn(1, K, device="cuda").bfloat16().expand(N, K)
    else:
        a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm_nt_2d(a, b, out)
    prof = torch.zeros(64 + 3 * 256 + 8, dtype=torch.int64, device="cuda")
    lib.lrp_debug_gemm_prof(prof.data_ptr())
    ops.gemm_nt_2d(a, b, out)
    torch.cuda.synchronize()
    lib.lrp_debug_gemm_prof(None)
    p = prof.cpu().numpy().astype(np.float64)
    nst = min(K // 32, 256)
    st = p[64: 64 + 3 * nst].reshape(nst, 3)
    print(f"--- cfg 26 M={M} N={N} K={K} {'hot' if hot else 'cold'}: prologue {p[1]-p[0]:.0f} cycles, loop {p[2]-p[1]:.0f} ({(p[2]-p[1])/(K//32):.0f} per 64-byte sub-step; ideal 1024)")
    issue = st[1:, 0] - st[:-1, 2]          # barrier passed -> MFMAs issued + fragments landed
    wl = st[:, 1] - st[:, 0]                # vmcnt wait
    wb = st[:, 2] - st[:, 1]                # barrier wait
    for name, v in (("MFMA issue + lgkm", issue), ("vmcnt wait", wl), ("barrier wait", wb)):
        print(f"    {name:18s} median {np.median(v):7.0f}  mean {v.mean():7.0f}  p90 {np.percentile(v, 90):7.0f}  max {v.max():7.0f}")
