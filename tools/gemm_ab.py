#!/usr/bin/env python3
"""A/B of lrp_gemm_nt builds on the GPU box (dev tool): every .so given on the command line is loaded side by side through
ctypes, checked against an fp32 torch reference on sampled rows, and timed in INTERLEAVED rounds (one process, same operands)
on the GEMM shapes of one Llama-3-8B layer at M = 8192 (B = 4 prompts x 2048).  torch.matmul (hipBLASLt) is timed beside them.

  python tools/gemm_ab.py [--rounds 3] [--iters 10] name=path.so [name=path.so ...]
A library built with -DPP_TIMELINE (name starting with "tl") additionally dumps the in-kernel interval timeline."""
import argparse
import ctypes
import os
import sys

import torch

SHAPES = [  # (M, N, K, what)
    (8192, 6144, 4096, "qkv fwd"), (8192, 4096, 4096, "o fwd/bwd"), (8192, 28672, 4096, "gate/up fwd"),
    (8192, 4096, 14336, "down fwd"), (8192, 14336, 4096, "down bwd"), (8192, 4096, 28672, "gate/up bwd"),
    (8192, 4096, 6144, "qkv bwd"),
]


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    fn = lib.lrp_gemm_nt
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_int] + [ctypes.c_int64] * 3 + \
        [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return fn


def call(fn, a, b, out, bias=None):
    M, K = a.shape
    N = b.shape[0]
    st = torch.cuda.current_stream().cuda_stream
    rc = fn(a.data_ptr(), b.data_ptr(), out.data_ptr(), bias.data_ptr() if bias is not None else None, M, N, K,
            a.stride(0), b.stride(0), out.stride(0), 1, 0, 0, 0, 1, 1 if out.dtype == torch.bfloat16 else 0, st)
    assert rc == 0, rc


def timeit(f, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def check(fn, name):
    """ragged + full shapes against fp32 matmul of the same bf16 operands"""
    torch.manual_seed(0)
    worst = 0.0
    for (M, N, K, odt, with_bias) in [(8192, 4096, 4096, torch.bfloat16, False), (8192, 6144, 128, torch.bfloat16, False),
                                      (8000, 6100, 192, torch.float32, True), (8192, 4096, 14336, torch.float32, False),
                                      (7937, 6144, 4096, torch.bfloat16, True), (8192, 8192, 320, torch.bfloat16, False)]:
        a = torch.randn(M, K, device="cuda").bfloat16()
        b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        bias = torch.randn(N, device="cuda").bfloat16() if with_bias else None
        out = torch.full((M, N), float("nan"), device="cuda", dtype=odt)
        call(fn, a, b, out, bias)
        torch.cuda.synchronize()
        rows = torch.cat([torch.arange(0, 300, device="cuda"), torch.randint(0, M, (212,), device="cuda"), torch.arange(M - 300, M, device="cuda")])
        ref = a[rows].float() @ b.float().T
        if bias is not None:
            ref = ref + bias.float()
        got = out[rows].float()
        err = float((got - ref).abs().max() / ref.abs().max())
        bad = int(torch.isnan(out).sum())
        tol = 1e-2 if odt == torch.bfloat16 else 2e-5 * max(1, K // 1024)
        flag = "ok" if (err < tol and bad == 0) else "FAIL"
        worst = max(worst, err)
        print(f"  check {name}: M={M} N={N} K={K} out={str(odt)[6:]} bias={with_bias}: err {err:.2e} nan {bad} {flag}", flush=True)
    return worst


def timeline_light(fn, name):
    """PP2_TIMELINE == 2: stamps [start, realtime, loop start, loop end, stores issued, stores retired, realtime] (realtime: 100 MHz)"""
    for (M, N, K) in [(8192, 28672, 4096), (8192, 4096, 14336), (8192, 4096, 4096), (8192, 14336, 4096)]:
        a = torch.randn(M, K, device="cuda").bfloat16()
        b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        tl = torch.zeros(3 * 2 * 192, device="cuda", dtype=torch.int64)
        for _ in range(5):
            call(fn, a, b, out, tl)
        torch.cuda.synchronize()
        t = tl.cpu().view(3, 2, 192)
        nph = K // 32
        for slot in range(3):
            for g in range(2):
                v = t[slot, g].tolist()
                if v[0] == 0:
                    continue
                clk = (v[5] - v[0]) / max(1, (v[6] - v[1])) * 0.1          # GHz
                loop = v[3] - v[2]
                print(f"  light {name} N={N} K={K} wg {slot} g{g}: prologue {v[2] - v[0]}  loop {loop} = {loop / nph:.0f} per phase (ideal 1024: "
                      f"{1024 * nph / loop * 100:.1f} % MFMA slots)  store issue {v[4] - v[3]}  drain {v[5] - v[4]}  total {v[5] - v[0]}"
                      f"  ({1024 * nph / (v[5] - v[0]) * 100:.1f} %)  shader clock {clk:.2f} GHz", flush=True)


def timeline(fn, name):
    """pp2 stamps: 0 = kernel start; then per phase: L end (before barrier), M start (after it), M end (before barrier), next L start"""
    for (M, N, K) in [(8192, 28672, 4096), (8192, 4096, 14336), (8192, 28672, 1024)]:
        a = torch.randn(M, K, device="cuda").bfloat16()
        b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        tl = torch.zeros(3 * 2 * 192, device="cuda", dtype=torch.int64)
        for _ in range(3):
            call(fn, a, b, out, tl)
        torch.cuda.synchronize()
        t = tl.cpu().view(3, 2, 192)
        nph = K // 32
        print(f"  timeline {name} N={N} K={K} ({nph} phases)")
        for slot in range(3):
            for g in range(2):
                v = t[slot, g].tolist()
                if v[0] == 0:
                    continue
                nfull = min(nph, (192 - 1) // 4)
                Lb, w1, Mb, w2 = [], [], [], []
                for ph in range(4, nfull - 1):            # steady state
                    s0 = v[4 * ph]                        # start of this phase's L (== stamp after the previous barrier)
                    a_, b_, c_, d_ = v[4 * ph + 1: 4 * ph + 5]
                    Lb.append(a_ - s0); w1.append(b_ - a_); Mb.append(c_ - b_); w2.append(d_ - c_)
                md = lambda x: sorted(x)[len(x) // 2] if x else -1      # noqa: E731
                mean = lambda x: sum(x) / max(1, len(x))              # noqa: E731
                per = mean(Lb) + mean(w1) + mean(Mb) + mean(w2)
                msg = (f"    wg {slot} group {g}: first L {v[1] - v[0]}  | per phase (median / mean): L busy {md(Lb)}/{mean(Lb):.0f}  wait {md(w1)}/{mean(w1):.0f}"
                       f"  M busy {md(Mb)}/{mean(Mb):.0f}  wait {md(w2)}/{mean(w2):.0f}  = {per:.0f} cycles per phase (2 intervals; ideal 1024)")
                if 4 * nph + 2 < 192:
                    e0 = v[4 * nph]
                    msg += f" | epilogue: store issue {v[4 * nph + 1] - e0} drain {v[4 * nph + 2] - v[4 * nph + 1]} | kernel span {v[4 * nph + 2] - v[0]}"
                print(msg, flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--m", type=int, default=8192, help="rows (B x S)")
    ap.add_argument("--siglip", action="store_true", help="the SigLIP-So400m tower's GEMM shapes (4 images x 4096 patches, K = 1152 / 4352) instead")
    a_ = ap.parse_args()
    print(torch.cuda.get_device_name(0), torch.version.hip, flush=True)
    libs = []
    for spec in a_.libs:
        name, path = spec.split("=", 1)
        libs.append((name, load(path)))
    for name, fn in libs:
        if name.startswith("tll"):
            timeline_light(fn, name)
        elif name.startswith("tl"):
            timeline(fn, name)
        elif not a_.no_check and not name.startswith("x_"):
            check(fn, name)
    libs = [(n, f) for (n, f) in libs if not n.startswith("tl")]
    res = {}
    shapes = [(a_.m, n_, k_, w_) for (_, n_, k_, w_) in SHAPES]
    if a_.siglip:
        shapes = [(16384, 3456, 1152, "qkv fwd"), (16384, 1152, 1152, "o fwd/bwd"), (16384, 4352, 1152, "fc1 fwd"), (16384, 1152, 4352, "fc2 fwd"),
                  (8192, 2560, 10240, "g3 down fwd"), (8192, 10240, 2560, "g3 gate fwd")]
    for (M, N, K, what) in shapes:
        a = torch.randn(M, K, device="cuda").bfloat16()
        b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        cands = [(n, (lambda f=f: call(f, a, b, out))) for (n, f) in libs] + [("hipblaslt", lambda: torch.matmul(a, b.T, out=out))]
        for n, f in cands:
            for _ in range(3):
                f()
        torch.cuda.synchronize()
        for r in range(a_.rounds):
            for n, f in cands:
                t = timeit(f, a_.iters)
                res.setdefault((M, N, K, what), {}).setdefault(n, []).append(fl / t / 1e12)
        line = f"M={M} N={N:6d} K={K:6d} {what:12s}: " + "  ".join(
            f"{n} {min(v):6.0f}-{max(v):6.0f}" for n, v in res[(M, N, K, what)].items())
        print(line, flush=True)
    # per-layer weighted time (one layer: all 7 shapes once, 'o' twice)
    names = list(next(iter(res.values())).keys())
    print("per-layer GEMM time (us), median TF/s per shape:")
    for n in names:
        tot = 0.0
        for (M, N, K, what), d in res.items():
            v = sorted(d[n])
            tf = v[len(v) // 2]
            tt = 2.0 * M * N * K / (tf * 1e12)
            tot += tt * (2 if what.startswith("o ") else 1)
        fl = sum(2.0 * M * N * K * (2 if w.startswith("o ") else 1) for (M, N, K, w) in res)
        print(f"  {n:12s} {tot * 1e6:9.1f} us   {fl / tot / 1e12:7.0f} TF/s", flush=True)
