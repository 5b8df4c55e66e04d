#!/usr/bin/env python3
"""GPU box: A/B of the Linear forward in its HBM-bound regime -- the one-launch weight-streaming kernel (ops.STREAM_FWD, lrp_linear_stream_fwd)
vs the split-K skinny path (lrp_gemm_skinny) -- same process, interleaved, three distinct weights rotated (beyond the Infinity Cache),
plus the dgrad (skinny NN / W-streaming small-M) beside it.  Usage: stream_ab.py [N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import lxt_amd.ops as ops  # noqa: E402


def timed(fn, n=21):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (14336, 4096)
    g = torch.Generator(device="cuda").manual_seed(0)
    nrot = 3 if N * K * 2 < 400e6 else 1
    Ws = [(torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).bfloat16() for _ in range(nrot)]
    print(f"# W [{N},{K}] bf16 x {nrot} rotated; us per launch (algorithmic TB/s)")
    for M in (1, 2, 4, 8, 16, 32, 64, 128, 160):
        x = torch.randn(M, K, generator=g, device="cuda").bfloat16()
        s = torch.randn(M, N, generator=g, device="cuda").bfloat16()
        z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        c = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
        by = 2 * (N * K + M * K + M * N)
        res = {}
        for name, flag in (("stream", True), ("skinny", False), ("stream2", True), ("skinny2", False)):
            ops.STREAM_FWD = flag
            res[name] = timed(lambda i: ops.linear_fwd(x, Ws[i % nrot], out=z))
        ops.STREAM_FWD = True
        dg = {}
        for name, flag in (("stream", True), ("old", False), ("stream2", True), ("old2", False)):
            ops.STREAM_DGRAD = flag
            dg[name] = timed(lambda i: ops.linear_dgrad(s, Ws[i % nrot], out=c))
        ops.STREAM_DGRAD = True
        print(f"M={M:4d} fwd stream {res['stream']:7.2f} / {res['stream2']:7.2f} ({by / res['stream2'] / 1e6:5.2f}) | skinny {res['skinny']:7.2f} / {res['skinny2']:7.2f} "
              f"({by / res['skinny2'] / 1e6:5.2f}) | dgrad stream {dg['stream']:7.2f} / {dg['stream2']:7.2f} ({by / dg['stream2'] / 1e6:5.2f}) | old "
              f"{dg['old']:7.2f} / {dg['old2']:7.2f} ({by / dg['old2'] / 1e6:5.2f})", flush=True)


if __name__ == "__main__":
    main()
