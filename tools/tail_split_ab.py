#!/usr/bin/env python3
"""Dev tool (GPU box): 257 ... 511-tile GEMMs with and without ops.TAIL_SPLIT (one full round of 256 tiles + the tail K-split over all CUs
against round 3's handling: two unbalanced rounds, or the whole problem split in two over K when K >= 8192).  Gemma-3-4B shapes, S = 2048 x 4."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import lxt_amd.ops as ops  # noqa: E402

SHAPES = [("o-proj fwd (NT)", False, 8192, 2560, 2048), ("down fwd (NT)", False, 8192, 2560, 10240), ("qkv dgrad (NN)", True, 8192, 2560, 4096),
          ("gate/up dgrad (NN)", True, 8192, 2560, 20480), ("o-proj fwd, 1 prompt x4 tiles (NT)", False, 2048, 10240, 2560)]


def bench(fn, n=30):
    for _ in range(5):
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
    for name, nn, M, N, K in SHAPES:
        # three operand sets in rotation (the 168 ... 336 MB activations must come from HBM as in the engine, not from the 256 MB Infinity Cache)
        sets = []
        for _ in range(3):
            a = torch.randn(M, K, generator=g).bfloat16().cuda()
            w = ((torch.randn(K, N, generator=g) if nn else torch.randn(N, K, generator=g)) * K ** -0.5).bfloat16().cuda()
            sets.append((a, w, torch.empty(M, N, dtype=torch.bfloat16, device="cuda")))
        it = [0]

        def fn():
            a, w, out = sets[it[0] % 3]
            it[0] += 1
            return ops.linear_dgrad(a, w, out=out) if nn else ops.linear_fwd(a, w, out=out)
        res = {}
        for flag in (False, True):
            ops.TAIL_SPLIT = flag
            res[flag] = bench(fn)
        fl = 2.0 * M * N * K
        print(f"{name:36s} M={M} N={N} K={K}: round-3 handling {res[False]:7.1f} us ({fl / res[False] * 1e-6:6.0f} TF/s) | tail split "
              f"{res[True]:7.1f} us ({fl / res[True] * 1e-6:6.0f} TF/s)", flush=True)
    ops.TAIL_SPLIT = True


if __name__ == "__main__":
    main()
