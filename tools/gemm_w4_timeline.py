"""dev: wait-point totals of the 4-wave GEMM built with -DW4_TIMELINE (tools/ab/dev_w4_timeline.so put in place of the library)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from lxt_amd import ops

names = ["mfma(s0, to lgkm wait)", "lgkm wait", "barrier0", "mfma(to vmcnt wait)", "vmcnt wait", "barrier1"]
for (M, N, K) in ((8192, 28672, 4096), (8192, 4096, 14336)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_nt_2d(a, b, out)
    torch.cuda.synchronize()
    w = out.view(torch.int32).view(M, N // 2)[::128].reshape(M // 128, N // 256, 128)[:, :, :].contiguous()
    w = out.view(torch.int32).view(M // 128, 128, N // 256, 128)[:, 0, :, :]          # row 0 of every wave-row block
    w = torch.cat([w[:, :, 0:8], w[:, :, 64:72]], 1).reshape(-1, 8).double().cpu()     # both waves of a tile column pair
    tot, nkt = w[:, 6].mean(), w[:, 7].mean()
    print(f"M={M} N={N} K={K}: loop {tot:.0f} cycles, {tot / nkt:.0f} per K tile of 64 (128 MFMAs/wave = 2048 ideal)  |  " +
          "  ".join(f"{n} {100 * w[:, i].mean() / tot:.1f}%" for i, n in enumerate(names)))
