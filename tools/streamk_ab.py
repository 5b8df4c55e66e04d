#!/usr/bin/env python3
"""stream-K (ops.STREAMK) against the round-4 dispatch (tail split / split-K slabs + reduce / partial last round) through ops.linear_fwd /
ops.linear_dgrad, per shape: the GEMMs of one Llama-3-8B layer at ONE prompt per step (M = 2048), Gemma-3-4B's N = 2560 shapes and the SigLIP
tower's at M = 16384.  Interleaved timing, engine operand layouts."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lxt_amd  # noqa: E402,F401
import lxt_amd.ops as ops  # noqa: E402

SHAPES = [("llama b1 qkv fwd", 2048, 6144, 4096, 0), ("llama b1 o fwd", 2048, 4096, 4096, 0), ("llama b1 gate/up fwd", 2048, 28672, 4096, 0),
          ("llama b1 down fwd", 2048, 4096, 14336, 0), ("llama b1 down bwd", 2048, 14336, 4096, 1), ("llama b1 gate/up bwd", 2048, 4096, 28672, 1),
          ("llama b1 qkv bwd", 2048, 4096, 6144, 1), ("llama b1 o bwd", 2048, 4096, 4096, 1),
          ("gemma o fwd", 8192, 2560, 2048, 0), ("gemma down fwd", 8192, 2560, 10240, 0), ("gemma gate/up bwd", 8192, 2560, 20480, 1),
          ("gemma qkv bwd", 8192, 2560, 4096, 1), ("siglip qkv fwd", 16384, 3456, 1152, 0), ("siglip fc1 fwd", 16384, 4352, 1152, 0),
          ("siglip fc2 fwd", 16384, 1152, 4352, 0), ("siglip o fwd", 16384, 1152, 1152, 0)]


def timeit(f, iters=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


bf = torch.bfloat16
print(f"{'shape':24s} {'M':>6s} {'N':>6s} {'K':>6s} {'tiles':>6s} {'streamK us':>11s} {'old us':>9s} {'TF/s new':>9s} {'TF/s old':>9s}")
for name, M, N, K, nn in SHAPES:
    a = torch.randn(M, K, device="cuda").to(bf)
    w = (torch.randn(K, N, device="cuda") if nn else torch.randn(N, K, device="cuda")).to(bf) * K ** -0.5
    out = torch.empty(M, N, device="cuda", dtype=bf)
    f = (lambda: ops.linear_dgrad(a, w, out=out)) if nn else (lambda: ops.linear_fwd(a, w, out=out))
    res = {}
    for rep in range(2):
        for flag in (True, False):
            ops.STREAMK = flag
            res.setdefault(flag, []).append(timeit(f))
    ops.STREAMK = True
    used = ops.streamk_ok(a, w, nn=bool(nn))
    tn, to = min(res[True]), min(res[False])
    fl = 2.0 * M * N * K
    print(f"{name:24s} {M:6d} {N:6d} {K:6d} {((M + 255) // 256) * ((N + 255) // 256):6d} {tn:11.1f} {to:9.1f} {fl / tn / 1e6:9.0f} {fl / to / 1e6:9.0f}  {'(stream-K taken)' if used else '(not taken)'}")
