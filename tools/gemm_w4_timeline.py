"""dev: wait-point / per-tile totals of the 4-wave GEMM (csrc/dev/gemm_w4.hip built with -DW4_TIMELINE and linked in place of the
product dispatch: tools/ab/dev_w4_timeline.so)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from lxt_amd import ops

names = ["mfma(to lgkm wait)", "lgkm wait", "barrier0", "mfma(to vmcnt wait)", "vmcnt wait", "barrier1"]
for (M, N, K) in ((8192, 28672, 4096), (8192, 4096, 14336), (8192, 4096, 4096)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(2):
        ops.gemm_nt_2d(a, b, out)
    torch.cuda.synchronize()
    w = out.view(torch.int32).view(M // 128, 128, N // 256, 128)[:, 0, :, :]          # row 0 of every wave-row block
    w = torch.cat([w[:, :, 0:12], w[:, :, 64:76]], 1).reshape(-1, 12).double().cpu()
    w = w[(w[:, 10] > 0) & (w[:, 10] < 100) & (w[:, 7] == w[:, 10] * (K // 64))]          # rows that hold a wave's counters (last tile of a workgroup)
    tiles, kern = w[:, 10].mean(), w[:, 11].mean()
    loop, pro, epi = w[:, 6].mean(), w[:, 8].mean(), w[:, 9].mean()
    print(f"M={M} N={N} K={K}: {len(w)} waves, {tiles:.1f} tiles each, kernel {kern:.0f} cycles | per tile: prologue {pro / tiles:.0f}  K loop {loop / tiles:.0f} "
          f"({loop / w[:, 7].mean():.0f} per K tile)  epilogue+store drain {epi / tiles:.0f}  | unaccounted {100 * (kern - loop - pro - epi) / kern:.1f}% | in loop: " +
          "  ".join(f"{n} {100 * w[:, i].mean() / loop:.1f}%" for i, n in enumerate(names)), flush=True)
