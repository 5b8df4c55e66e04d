#!/usr/bin/env python3
"""The Linear eps-rule in its HBM-bound regime, M = 1 ... 256 rows (dev tool; bench.py carries the judged table): forward z = x W^T and
redistribution c = s W (W as stored), through (a) the W-streaming small-M kernels (M <= 16), (b) the split-K skinny path of the
ping-pong GEMM (lrp_gemm_skinny, NT / NN), (c) torch.matmul for reference.  Algorithmic bytes = 2 (N K + M K + M N)."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import lxt_amd.ops as ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    print(torch.cuda.get_device_name(0))
    for (N, K, what) in [(14336, 4096, "gate/up-sized"), (4096, 14336, "down-sized"), (128256, 4096, "LM head")]:
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        for M in (1, 2, 4, 8, 16, 32, 64, 128, 160, 256):
            x = torch.randn(M, K, device="cuda").bfloat16()
            s = torch.randn(M, N, device="cuda").bfloat16()
            z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            c = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
            by = 2.0 * (N * K + M * K + M * N)
            row = f"{what:14s} M={M:3d}: "
            if M <= ops.SMALLM_MAX:
                tf = timeit(lambda: ops.linear_smallm_fwd(x, W, out=z))
                tb = timeit(lambda: ops.linear_smallm_dgrad(s, W, out=c))
                row += f"smallm fwd {by / tf / 1e9:6.0f}  dgrad {by / tb / 1e9:6.0f} GB/s | "
            tf = timeit(lambda: ops.gemm_skinny(x, W, z, nn=False))
            tb = timeit(lambda: ops.gemm_skinny(s, W, c, nn=True))
            tm = timeit(lambda: torch.matmul(x, W.T, out=z))
            tmb = timeit(lambda: torch.matmul(s, W, out=c))
            row += (f"skinny fwd {by / tf / 1e9:6.0f} ({tf * 1e6:6.1f} us)  dgrad {by / tb / 1e9:6.0f} ({tb * 1e6:6.1f} us) GB/s | "
                    f"torch fwd {by / tm / 1e9:6.0f}  dgrad {by / tmb / 1e9:6.0f}")
            print(row, flush=True)


if __name__ == "__main__":
    main()
