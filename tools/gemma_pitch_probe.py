#!/usr/bin/env python3
"""dev (GPU box): stored-weight row pitch for the Gemma-3-4B layer shapes (H 2560, I 10240, 8 + 4 heads of 256) and SigLIP (H 1152, I 4352):
NN (dgrad) and NT (forward) forms of the ping-pong GEMM through the product dispatch, M = 8192 / 16384, W stored at pitch cols + pad."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import lxt_amd.ops as ops

def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (M, rows, cols, what) in ((8192, 20480, 2560, "gemma gate/up"), (8192, 4096, 2560, "gemma qkv"), (8192, 2560, 2048, "gemma o"), (8192, 2560, 10240, "gemma down"),
                              (16384, 3456, 1152, "siglip qkv"), (16384, 4352, 1152, "siglip fc1"), (16384, 1152, 4352, "siglip fc2")):
    for form in ("dgrad", "fwd"):
        line = f"{what:14s} {form:5s} W[{rows},{cols}]:"
        for pad in (0, 64, 128, 192):
            Wb = (torch.randn(rows, cols + pad, device="cuda") * cols ** -0.5).bfloat16()
            W = Wb[:, :cols]
            if form == "dgrad":
                a = torch.randn(M, rows, device="cuda").bfloat16()
                f = lambda: ops.linear_dgrad(a, W)
            else:
                a = torch.randn(M, cols, device="cuda").bfloat16()
                f = lambda: ops.linear_fwd(a, W)
            s = t(f)
            line += f"  pad {pad}: {s * 1e6:7.1f} us ({2.0 * M * rows * cols / s / 1e12:5.0f} TF)"
            del Wb, W, a
        print(line, flush=True)
