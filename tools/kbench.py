#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box (dev tool; bench.py is the judged benchmark).
Times the C-ABI kernels at Llama-3-8B shapes with HIP events on torch's current stream."""
import argparse
import os
import sys
import time

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


def bench_gemm(dtype, shapes):
    for (M, N, K) in shapes:
        a = torch.randn(M, K, device="cuda").to(dtype)
        b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dtype)
        out = torch.empty(M, N, device="cuda", dtype=dtype)
        t = timeit(lambda: ops.gemm_nt_2d(a, b, out))
        tt = timeit(lambda: torch.matmul(a, b.T, out=out))
        print(f"gemm {str(dtype)[6:]:8s} M={M:5d} N={N:6d} K={K:6d}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TF/s"
              f"   | hipBLASLt (torch) {tt*1e6:9.1f} us {2*M*N*K/tt/1e12:8.1f} TF/s", flush=True)


def bench_gemm_pad(dtype, shapes, pads=(0, 64, 128, 192, 256, 512)):
    """leading-dimension sweep: rows of A, B (and C) padded by `pad` elements -- power-of-two row pitches map the
    same K slice of every row to the same L2/HBM channel (channel camping)"""
    for (M, N, K) in shapes:
        for pad in pads:
            a = torch.randn(M, K + pad, device="cuda").to(dtype)[:, :K]
            b = (torch.randn(N, K + pad, device="cuda") * K ** -0.5).to(dtype)[:, :K]
            out = torch.empty(M, N + pad, device="cuda", dtype=dtype)[:, :N]
            t = timeit(lambda: ops.gemm_nt_2d(a, b, out))
            print(f"gemm-pad {str(dtype)[6:]:8s} M={M:5d} N={N:6d} K={K:6d} pad={pad:4d}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TF/s", flush=True)


def bench_gemm_hot(dtype, shapes):
    """cache-hot variant: every row of A and B aliases row 0 (row stride 0) -> all global loads hit L1/L2;
    isolates the in-CU efficiency (LDS + MFMA + barriers) from the memory system"""
    for (M, N, K) in shapes:
        a = torch.randn(1, K, device="cuda").to(dtype).expand(M, K)
        b = (torch.randn(1, K, device="cuda") * K ** -0.5).to(dtype).expand(N, K)
        out = torch.empty(M, N, device="cuda", dtype=dtype)
        t = timeit(lambda: ops.gemm_nt_2d(a, b, out))
        print(f"gemm-hot {str(dtype)[6:]:8s} M={M:5d} N={N:6d} K={K:6d}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TF/s", flush=True)


def bench_attn(dtype, B, S, Hq, Hkv, d):
    q = torch.randn(B * S, Hq * d, device="cuda").to(dtype)
    k = torch.randn(B * S, Hkv * d, device="cuda").to(dtype)
    v = torch.randn(B * S, Hkv * d, device="cuda").to(dtype)
    v_t, k_t, q_t = ops.transpose_heads(v, B, S, Hkv, d), ops.transpose_heads(k, B, S, Hkv, d), ops.transpose_heads(q, B, S, Hq, d)
    o = torch.empty_like(q)
    lse = torch.empty(B, Hq, S, device="cuda")
    sc = d ** -0.5
    fl = 2 * 2 * B * Hq * S * S * d / 2
    t = timeit(lambda: ops.attn_fwd(q, k, v, v_t, o, lse, B, S, Hq, Hkv, d, sc, True, 0))
    print(f"attn fwd  {str(dtype)[6:]:8s} B={B} S={S} Hq={Hq} d={d}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s (causal flops)", flush=True)
    Go = torch.randn_like(q)
    Gho, D = torch.empty_like(q), torch.empty(B, Hq, S, device="cuda")
    t = timeit(lambda: ops.attn_bwd_prep(Go, o, Gho, D, B, S, Hq, d, 1e-6, 0.5))
    print(f"attn prep: {t*1e6:9.1f} us  {3*q.numel()*q.element_size()/t/1e9:7.1f} GB/s")
    Gho_t = ops.transpose_heads(Gho, B, S, Hq, d)
    t = timeit(lambda: ops.transpose_heads(q, B, S, Hq, d, out=q_t))
    print(f"transpose_heads q: {t*1e6:9.1f} us  {2*q.numel()*q.element_size()/t/1e9:7.1f} GB/s")
    dq = torch.empty_like(q)
    dk, dv = torch.empty_like(q), torch.empty_like(q)
    for name, e in (("efficient", 0.0), ("explicit", 1e-8)):
        t = timeit(lambda: ops.attn_bwd_dq(q, k, v, k_t, Gho, lse, D, dq, B, S, Hq, Hkv, d, sc, e, e))
        print(f"attn dq   {name}: {t*1e6:9.1f} us  {1.5*fl/t/1e12:7.1f} TF/s", flush=True)
        t = timeit(lambda: ops.attn_bwd_dkv(q, k, v, q_t, Gho, Gho_t, lse, D, dk, dv, B, S, Hq, Hkv, d, sc, e, e))
        print(f"attn dkv  {name}: {t*1e6:9.1f} us  {2*fl/t/1e12:7.1f} TF/s", flush=True)


def bench_row(dtype, M, H, I):
    h, br = torch.randn(M, H, device="cuda").to(dtype), torch.randn(M, H, device="cuda").to(dtype)
    w = torch.randn(H, device="cuda").to(dtype)
    hs, y, rstd = torch.empty_like(h), torch.empty_like(h), torch.empty(M, device="cuda")
    es = h.element_size()
    t = timeit(lambda: ops.add_rmsnorm_fwd(h, br, w, 1e-5, hsum_out=hs, y=y, rstd=rstd))
    print(f"add_rmsnorm_fwd [{M},{H}]: {t*1e6:8.1f} us  {4*M*H*es/t/1e9:7.1f} GB/s")
    Gs, A = torch.empty_like(h), torch.empty_like(h)
    t = timeit(lambda: ops.rmsnorm_bwd_add2(h, br, w, rstd, hs, br, Gs, A, None, 0.0, 1e-8, 1e-8))
    print(f"rmsnorm_bwd_add2 [{M},{H}]: {t*1e6:8.1f} us  {6*M*H*es/t/1e9:7.1f} GB/s")
    gu = torch.randn(M, 2 * I, device="cuda").to(dtype)
    m = torch.empty(M, I, device="cuda", dtype=dtype)
    t = timeit(lambda: ops.gated_act_fwd(gu[:, :I], gu[:, I:], m))
    print(f"gated_act_fwd [{M},{I}]: {t*1e6:8.1f} us  {3*M*I*es/t/1e9:7.1f} GB/s")
    Agu = torch.empty_like(gu)
    t = timeit(lambda: ops.gated_act_bwd(m, gu[:, :I], gu[:, I:], Agu[:, :I], Agu[:, I:], 1e-8, 1e-8))
    print(f"gated_act_bwd [{M},{I}]: {t*1e6:8.1f} us  {5*M*I*es/t/1e9:7.1f} GB/s", flush=True)


def bench_smallm_stream(dtype, M, N, K):
    """W-streaming forward and dgrad (lrp_linear_smallm_fwd / _dgrad): algorithmic bytes = sizeof * N * K (+ M-row operands)"""
    x, W = torch.randn(M, K, device="cuda").to(dtype), (torch.randn(N, K, device="cuda") * K ** -0.5).to(dtype)
    g = torch.randn(M, N, device="cuda").to(dtype)
    es = x.element_size()
    z = ops.linear_smallm_fwd(x, W)
    t = timeit(lambda: ops.linear_smallm_fwd(x, W, out=z))
    print(f"smallm fwd   {str(dtype)[6:]} M={M:2d} N={N} K={K}: {t*1e6:8.1f} us  {es*(N*K+M*K+M*N)/t/1e9:7.1f} GB/s algorithmic", flush=True)
    out = ops.linear_smallm_dgrad(g, W, z=z, eps=1e-6)
    t = timeit(lambda: ops.linear_smallm_dgrad(g, W, z=z, eps=1e-6, out=out))
    print(f"smallm dgrad {str(dtype)[6:]} M={M:2d} N={N} K={K}: {t*1e6:8.1f} us  {es*(N*K+M*K+2*M*N)/t/1e9:7.1f} GB/s algorithmic", flush=True)
    tt = timeit(lambda: torch.matmul(x, W.T))
    print(f"   (torch.matmul forward, same shape: {tt*1e6:8.1f} us  {es*N*K/tt/1e9:7.1f} GB/s)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gemm,attn,row,stream")
    a = ap.parse_args()
    print(torch.cuda.get_device_name(0), torch.version.hip)
    L = [(2048, 6144, 4096), (2048, 4096, 4096), (2048, 28672, 4096), (2048, 4096, 14336), (2048, 14336, 4096), (2048, 4096, 28672),
         (2048, 4096, 6144), (8192, 4096, 4096), (8192, 28672, 4096)]
    if "hot" in a.what:
        bench_gemm(torch.bfloat16, [(8192, 4096, 4096)])
        bench_gemm_hot(torch.bfloat16, [(8192, 4096, 4096), (8192, 28672, 4096), (8192, 4096, 14336)])
    if "pad" in a.what:
        bench_gemm_pad(torch.bfloat16, [(8192, 4096, 4096), (8192, 28672, 4096), (8192, 4096, 14336)])
    if "onegemm" in a.what:
        bench_gemm(torch.bfloat16, [(8192, 4096, 4096), (8192, 28672, 4096), (8192, 4096, 14336)])
    elif "gemm" in a.what:
        bench_gemm(torch.bfloat16, L)
        bench_gemm(torch.float32, [(2048, 4096, 4096), (2048, 14336, 4096)])
    if "attnb" in a.what:
        bench_attn(torch.bfloat16, 4, 2048, 32, 8, 128)
    elif "attn" in a.what:
        bench_attn(torch.bfloat16, 1, 2048, 32, 8, 128)
        bench_attn(torch.bfloat16, 4, 2048, 32, 8, 128)
        bench_attn(torch.float32, 1, 2048, 32, 8, 128)
    if "row" in a.what:
        bench_row(torch.bfloat16, 2048, 4096, 14336)
    if "stream" in a.what:
        for M in (1, 2, 4, 8, 16):
            bench_smallm_stream(torch.bfloat16, M, 14336, 4096)
        for M in (1, 4, 16):
            bench_smallm_stream(torch.bfloat16, M, 4096, 14336)
        bench_smallm_stream(torch.bfloat16, 4, 128256, 4096)
        bench_smallm_stream(torch.float32, 1, 768, 768)
