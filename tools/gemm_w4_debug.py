"""dev: correctness of the 4-wave GEMM over a grid of shapes (library built with -DLRP_W4_MIN_TILES=1)"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from lxt_amd import ops
torch.manual_seed(0)
for (M, N, K) in ((256, 256, 128), (256, 256, 192), (256, 256, 256), (256, 256, 1024), (512, 256, 192), (256, 512, 192), (512, 512, 256),
                  (4096, 4096, 192), (4096, 4608, 256), (8192, 4096, 1024), (8192, 8192, 192), (300, 500, 192), (8192, 4096, 14336)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt_2d(a, b, out)
    ref = a.float() @ b.float().T
    err = (out.float() - ref).abs()
    # which 256x256 output tiles are wrong, and is the error a missing K slice?
    tm, tn = (M + 255) // 256, (N + 255) // 256
    e = torch.nn.functional.pad(err, (0, tn * 256 - N, 0, tm * 256 - M)).view(tm, 256, tn, 256).amax((1, 3))
    bad = (e > 0.05 * ref.abs().max()).nonzero().tolist()
    print(f"M={M} N={N} K={K}: max err {float(err.max()):.3f} (ref max {float(ref.abs().max()):.2f})  bad tiles {len(bad)}/{tm*tn} {bad[:6]}", flush=True)
    if bad and K <= 1024:
        i, j = bad[0]
        sub = out[i*256:(i+1)*256, j*256:(j+1)*256].float()
        for kt in range(K // 64):
            part = a[i*256:(i+1)*256, kt*64:(kt+1)*64].float() @ b[j*256:(j+1)*256, kt*64:(kt+1)*64].float().T
            r2 = ref[i*256:(i+1)*256, j*256:(j+1)*256] - part
            print(f"    without K tile {kt}: err {float((sub - r2).abs().max()):.3f}", flush=True)
