#!/usr/bin/env python3
"""Dev tool (GPU box): plain ping-pong GEMM (NT and NN), M = 8192, N = 4096 (512 tiles = 2 rounds of the chip), K swept: the intercept of
T(K) = rounds x (a + b K) is the per-tile prologue + epilogue that a persistent kernel with cross-tile prefetch could hide."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import lxt_amd.ops as ops  # noqa: E402


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    g = torch.Generator().manual_seed(0)
    M, N = 8192, 4096
    for nn in (False, True):
        rows = []
        for K in (512, 1024, 2048, 4096, 8192, 14336):
            sets = []
            for _ in range(3):
                a = torch.randn(M, K, generator=g).bfloat16().cuda()
                w = ((torch.randn(K, N, generator=g) if nn else torch.randn(N, K, generator=g)) * K ** -0.5).bfloat16().cuda()
                sets.append((a, w, torch.empty(M, N, dtype=torch.bfloat16, device="cuda")))
            it = [0]

            def fn():
                a, w, o = sets[it[0] % 3]
                it[0] += 1
                return ops.gemm_nn_2d(a, w, o) if nn else ops.gemm_nt_2d(a, w, o)
            t = bench(fn)
            rows.append((K, t))
            print(f"{'NN' if nn else 'NT'} M={M} N={N} K={K:6d}: {t:8.1f} us  {2.0 * M * N * K / t * 1e-6:7.0f} TF/s", flush=True)
        (k1, t1), (k2, t2) = rows[-3], rows[-1]
        b = (t2 - t1) / (k2 - k1)
        a0 = t1 - b * k1
        print(f"   fit over K = {k1} .. {k2}: T = {a0:.1f} us + {b * 1e3:.2f} us per 1000 K  (2 rounds: {a0 / 2:.1f} us per tile outside the K loop; "
              f"K-loop rate {2.0 * M * N / b * 1e-6:.0f} TF/s)", flush=True)


if __name__ == "__main__":
    main()
