#!/usr/bin/env python3
"""dev (GPU box): does the row pitch of the STORED weight matter for the NN (dgrad) form?  c[M, Kout] = a[M, N] . W[N, Kout]: the B operand is
read as 512-byte segments of consecutive contraction rows, i.e. with the weight's row pitch as the stride.  Times lrp_gemm_nn on the three dgrad
shapes of a Llama-3-8B layer with W stored at pitch Kout + pad."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import lxt_amd.ops as ops

M = 8192
def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (N, Kout, what) in ((28672, 4096, "gate/up dgrad"), (6144, 4096, "qkv dgrad"), (4096, 4096, "o dgrad"), (4096, 14336, "down dgrad (plain)")):
    apad = 64 if (N * 2) % 4096 == 0 and N * 2 >= 16384 else 0
    a = torch.randn(M, N + apad, device="cuda").bfloat16()[:, :N]
    out = torch.empty(M, Kout, device="cuda", dtype=torch.bfloat16)
    line = f"{what:20s} M={M} N(contraction)={N} Kout={Kout}:"
    for pad in (0, 64, 128, 256, 2048):
        Wb = (torch.randn(N, Kout + pad, device="cuda") * N ** -0.5).bfloat16()
        W = Wb[:, :Kout]
        s = t(lambda: ops.gemm_nn_2d(a, W, out))
        line += f"  pad {pad}: {s * 1e6:7.1f} us ({2.0 * M * N * Kout / s / 1e12:5.0f} TF)"
        del Wb, W
    print(line, flush=True)
