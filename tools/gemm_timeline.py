"""dev: in-kernel timeline of the persistent 256x256 GEMM (LRP_GEMM_TILE=25): where a tile's time goes"""
import os, sys, torch
os.environ["LRP_GEMM_TILE"] = "25"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lxt_amd.ops as ops
from lxt_amd._lib import lib
import numpy as np
for (M, N, K, hot) in [(8192, 4096, 4096, 0), (8192, 4096, 4096, 1), (8192, 28672, 4096, 0)]:
    if hot:
        a = torch.randn(1, K, device="cuda").bfloat16().expand(M, K); b = torch.randn(1, K, device="cuda").bfloat16().expand(N, K)
    else:
        a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm_nt_2d(a, b, out)
    G = 256
    prof = torch.zeros(G * 16 + 512, dtype=torch.int64, device="cuda")
    lib.lrp_debug_gemm_prof(prof.data_ptr())
    ops.gemm_nt_2d(a, b, out)
    torch.cuda.synchronize()
    lib.lrp_debug_gemm_prof(None)
    p = prof.cpu().numpy()
    ev = p[: G * 16].reshape(G, 16).astype(np.float64)
    t0 = ev[:, 0].min()
    ev = np.where(ev > 0, ev - t0, np.nan)
    ntile = (M // 256) * (N // 256)
    rounds = ntile // G
    print(f"--- M={M} N={N} K={K} {'hot' if hot else 'cold'}: {rounds} tiles per workgroup; shader-clock cycles, median over workgroups (min..max)")
    names = ["start"] + [f"t{r}:{n}" for r in range(5) for n in ("landed", "loop_done", "stored")]
    prev = None
    for i in range(min(16, 1 + 3 * rounds)):
        col = ev[:, i]
        med = np.nanmedian(col)
        d = "" if prev is None else f"  (+{med - prev:9.0f})"
        print(f"  {names[i]:14s} {med:10.0f}  [{np.nanmin(col):9.0f} .. {np.nanmax(col):9.0f}]{d}")
        prev = med
    steps = p[G * 16: G * 16 + K // 64].astype(np.float64)
    ds = np.diff(steps)
    print(f"  per K step (wg 0, tile 0): first 6 {ds[:6].astype(int).tolist()}  median {np.median(ds):.0f}  p90 {np.percentile(ds, 90):.0f}  max {ds.max():.0f}  (ideal MFMA-bound: 2048)")
