#!/usr/bin/env python3
"""GPU box: timings of the three attention kernels at a given shape (default: Gemma-3-4B, B = 4, S = 2048, 8 / 4 heads of d = 256, causal,
window 0 and 1024; then Llama-3-8B's 32 / 8 heads of d = 128).  FLOPs: 2 contractions forward, 5 backward (7 algorithmic), causal half."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import lxt_amd.ops as ops  # noqa: E402


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


def bench(B, S, Hq, Hkv, d, window, mode, causal=True, as_intervals=False):
    g = torch.Generator(device="cuda").manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g, device="cuda").bfloat16()   # noqa: E731
    q, k, v, Go = rn(B * S, Hq * d), rn(B * S, Hkv * d), rn(B * S, Hkv * d), rn(B * S, Hq * d)
    need_t = ops.attn_needs_transposed(q, d)
    v_t = ops.transpose_heads(v, B, S, Hkv, d) if need_t else None
    o, lse = torch.empty_like(q), torch.empty(B, Hq, S, device="cuda")
    sc = d ** -0.5
    e = 1e-8 if mode == "explicit" else 0.0
    vis = float(S) * S if not causal else ((S * (S + 1) / 2) if window <= 0 else sum(min(i + 1, window) for i in range(S)))
    unit = 2.0 * vis * d * Hq * B                        # one contraction over the visible scores
    row_iv, tag = None, ""
    if as_intervals:
        # the SAME causal (+ window) mask, expressed ONLY as per-row key intervals with a bidirectional 256-token image block (what the Gemma-3
        # image + text driver hands over: causal = 0, window = 0): the kernels must find their tile ranges in the intervals
        i = torch.arange(S)
        lo = (i - window + 1).clamp_min(0) if window > 0 else torch.zeros(S, dtype=torch.long)
        hi = i + 1
        hi[64:320] = 320
        lo[64:320] = lo[64]
        row_iv = (lo.int().repeat(B, 1).cuda().contiguous(), hi.int().repeat(B, 1).cuda().contiguous())
        causal, window, tag = False, 0, " [mask as intervals]"
    tf = timeit(lambda: ops.attn_fwd(q, k, v, v_t, o, lse, B, S, Hq, Hkv, d, sc, causal, window, row_iv=row_iv))
    Gho, D = torch.empty_like(q), torch.empty(B, Hq, S, device="cuda")
    ops.attn_bwd_prep(Go, o, Gho, D, B, S, Hq, d, 1e-6 if e else 0.0, 0.5)
    k_t = ops.transpose_heads(k, B, S, Hkv, d) if need_t else None
    q_t = ops.transpose_heads(q, B, S, Hq, d) if need_t else None
    Gho_t = ops.transpose_heads(Gho, B, S, Hq, d) if need_t else None
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    tq = timeit(lambda: ops.attn_bwd_dq(q, k, v, k_t, Gho, lse, D, dq, B, S, Hq, Hkv, d, sc, e, e, causal, window, row_iv=row_iv))
    tk = timeit(lambda: ops.attn_bwd_dkv(q, k, v, q_t, Gho, Gho_t, lse, D, dk, dv, B, S, Hq, Hkv, d, sc, e, e, causal, window, row_iv=row_iv))
    tt = 0.0
    if need_t:
        tt = timeit(lambda: (ops.transpose_heads(v, B, S, Hkv, d), ops.transpose_heads(k, B, S, Hkv, d), ops.transpose_heads(q, B, S, Hq, d),
                             ops.transpose_heads(Gho, B, S, Hq, d)))
    tot = tf + tq + tk + tt
    print(f"B={B} S={S} {Hq}/{Hkv} heads d={d} window={window} {'causal' if causal else 'full'} {mode}{tag}: fwd {tf * 1e6:7.1f} us ({2 * unit / tf / 1e12:6.1f} TF/s) | dQ {tq * 1e6:7.1f} us "
          f"({2 * unit / tq / 1e12:6.1f}) | dK/dV {tk * 1e6:7.1f} us ({3 * unit / tk / 1e12:6.1f}) | head transposes {tt * 1e6:6.1f} us | composite "
          f"{7 * unit / tot / 1e12:6.1f} TF/s = {7 * unit / tot / 2.5e15:.3f} of peak, {tot * 1e6:7.1f} us per layer", flush=True)


if __name__ == "__main__":
    for w in (0, 1024):
        bench(4, 2048, 8, 4, 256, w, "efficient")
        bench(4, 2048, 8, 4, 256, w, "efficient", as_intervals=True)
    bench(4, 2048, 32, 8, 128, 0, "efficient")
    bench(4, 2048, 32, 8, 128, 0, "efficient", as_intervals=True)
    bench(4, 2048, 32, 8, 128, 0, "explicit")
    bench(4, 2048, 32, 8, 64, 0, "efficient")                       # Llama-3.2-1B-like heads of d = 64
    bench(64, 128, 12, 12, 64, 0, "efficient", causal=False)        # BERT-base, batch 64
    for d in (96, 128):                                              # SigLIP tower: 4096 patches, 16 heads of d = 72 padded to 96 / 128
        bench(4, 4096, 16, 16, d, 0, "efficient", causal=False)
    bench(64, 128, 12, 12, 64, 0, "efficient")
