#!/usr/bin/env python3
"""A/B of liblrp_hip.so builds on the GPU box (dev tool) -- and the gate of VERDICT r5 item 1(b): the ping-pong GEMM against hipBLASLt.
Every .so named on the command line is loaded side by side through ctypes and timed in INTERLEAVED rounds (one process, same operands, the
ENGINE's operand layouts: weight / activation row pitches off the 4-KiB grid) on the seven GEMMs of one Llama-3-8B layer at M = 8192
(B = 4 prompts x 2048), in the form the engine launches them:
    forward  z = x W^T        lrp_gemm_nt      (+ the K1n forms lrp_gemm_nt_rs / lrp_gemm_res_ssq)
    dgrad    c = s W          lrp_gemm_nn      (+ lrp_gemm_nn_rs / lrp_gemm_nn_rs_res), W [out, in] as stored
torch.matmul (hipBLASLt) is timed beside them on the same operands (NT and NN).

  python tools/gemm_ab.py [--rounds 3] [--iters 10] [--gemma] name=path.so [name=path.so ...]"""
import argparse
import ctypes
import os

import torch

I64, I32, VP = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
LLAMA = [  # (what, kind, M, N, K): kind nt -> W [N, K]; nn -> W [K, N] (a stored weight read as the dgrad operand)
    ("qkv fwd", "nt_rs", 8192, 6144, 4096), ("o fwd", "res_ssq", 8192, 4096, 4096), ("gate/up fwd", "nt", 8192, 28672, 4096),
    ("down fwd", "res_ssq", 8192, 4096, 14336), ("down bwd", "nn", 8192, 14336, 4096), ("gate/up bwd", "nn_rs_res", 8192, 4096, 28672),
    ("o bwd", "nn_rs", 8192, 4096, 4096), ("qkv bwd", "nn_rs_res", 8192, 4096, 6144),
]
GEMMA = [("g3 down fwd", "nt", 8192, 2560, 10240), ("g3 gate/up fwd", "nt", 8192, 20480, 2560), ("g3 down bwd", "nn", 8192, 10240, 2560),
         ("g3 gate/up bwd", "nn", 8192, 2560, 20480), ("siglip qkv", "nt", 16384, 3456, 1152), ("siglip fc1", "nt", 16384, 4352, 1152),
         ("siglip fc2", "nt", 16384, 1152, 4352)]


def pad_cols(cols, es=2):
    nb = cols * es
    return 128 // es if (nb >= 16384 and nb % 4096 == 0) else (0)


def wpad(cols, rows, es=2):
    nb = cols * es
    return 256 // es if (nb % 1024 == 0 and rows * nb >= (32 << 20)) else 0


class Lib:
    def __init__(self, path):
        L = self.L = ctypes.CDLL(os.path.abspath(path))
        L.lrp_gemm_nt.argtypes = [VP] * 4 + [I32] * 3 + [I64] * 3 + [I32] + [I64] * 3 + [I32, I32, VP]
        L.lrp_gemm_nn.argtypes = [VP] * 4 + [I32] * 3 + [I64] * 3 + [I32, I32, VP]
        L.lrp_gemm_nt_rs.argtypes = [VP] * 4 + [I32] * 3 + [I64] * 3 + [I32, VP]
        L.lrp_gemm_nn_rs.argtypes = [VP] * 4 + [I32] * 3 + [I64] * 3 + [I32, VP]
        self.v7 = L.lrp_version() >= 7                # ABI 7: lrp_gemm_res_ssq takes (raw, ldraw) ahead of dtype
        L.lrp_gemm_res_ssq.argtypes = [VP] * 5 + [I32] * 3 + [I64] * 5 + ([VP, I64] if self.v7 else []) + [I32, VP]
        L.lrp_gemm_nn_rs_res.argtypes = [VP] * 5 + [I32] * 3 + [I64] * 4 + [I32, VP]

    def run(self, kind, a, w, out, rs, res, ssq):
        M, K = a.shape
        N = out.shape[1]
        st = torch.cuda.current_stream().cuda_stream
        p = lambda t: t.data_ptr()      # noqa: E731
        L = self.L
        if kind == "nt":
            rc = L.lrp_gemm_nt(p(a), p(w), p(out), None, M, N, K, a.stride(0), w.stride(0), out.stride(0), 1, 0, 0, 0, 1, 1, st)
        elif kind == "nn":
            rc = L.lrp_gemm_nn(p(a), p(w), p(out), None, M, N, K, a.stride(0), w.stride(0), out.stride(0), 1, 1, st)
        elif kind == "nt_rs":
            rc = L.lrp_gemm_nt_rs(p(a), p(w), p(rs), p(out), M, N, K, a.stride(0), w.stride(0), out.stride(0), 1, st)
        elif kind == "nn_rs":
            rc = L.lrp_gemm_nn_rs(p(a), p(w), p(rs), p(out), M, N, K, a.stride(0), w.stride(0), out.stride(0), 1, st)
        elif kind == "res_ssq":
            rc = L.lrp_gemm_res_ssq(p(a), p(w), p(res), p(out), p(ssq), M, N, K, a.stride(0), w.stride(0), res.stride(0), out.stride(0), ssq.stride(0),
                                    *((None, 0) if self.v7 else ()), 1, st)
        else:
            rc = L.lrp_gemm_nn_rs_res(p(a), p(w), p(rs), p(res), p(out), M, N, K, a.stride(0), w.stride(0), res.stride(0), out.stride(0), 1, st)
        assert rc == 0, (kind, rc)


def timeit(f, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--gemma", action="store_true", help="the Gemma-3-4B / SigLIP shapes (tile counts that are no multiple of 256) instead")
    a_ = ap.parse_args()
    print(torch.cuda.get_device_name(0), torch.version.hip, flush=True)
    libs = [(s.split("=", 1)[0], Lib(s.split("=", 1)[1])) for s in a_.libs]
    bf = torch.bfloat16
    res_tab = {}
    for (what, kind, M, N, K) in (GEMMA if a_.gemma else LLAMA):
        nn = kind.startswith("nn")
        a = torch.randn(M, K + pad_cols(K), device="cuda").to(bf)[:, :K]
        if nn:      # stored weight [K rows = out features, N cols = in features]
            w = (torch.randn(K, N + wpad(N, K), device="cuda") * K ** -0.5).to(bf)[:, :N]
        else:
            w = (torch.randn(N, K + max(pad_cols(K), wpad(K, N)), device="cuda") * K ** -0.5).to(bf)[:, :K]
        out = torch.empty(M, N, device="cuda", dtype=bf)
        rs = torch.rand(M, device="cuda") + 0.5
        res = torch.randn(M, N, device="cuda").to(bf)
        ssq = torch.empty(max(N // 64, 1), M, device="cuda")
        k_eff = kind if (N % 256 == 0 or kind in ("nt", "nn")) else ("nn" if nn else "nt")
        fl = 2.0 * M * N * K
        cands = [(n, (lambda L=L: L.run(k_eff, a, w, out, rs, res, ssq))) for (n, L) in libs]
        cands.append(("hipblaslt", (lambda: torch.matmul(a, w, out=out)) if nn else (lambda: torch.matmul(a, w.T, out=out))))
        # correctness of every build against fp32 on sampled rows (plain product; the epilogue forms are covered by tests/)
        rows = torch.randint(0, M, (64,), device="cuda")
        ref = a[rows].float() @ (w.float() if nn else w.float().T)
        for n, L in libs:
            L.run("nn" if nn else "nt", a, w, out, rs, res, ssq)
            err = float((out[rows].float() - ref).abs().max() / ref.abs().max())
            assert err < 1e-2, (n, what, err)
        for n, f in cands:
            for _ in range(3):
                f()
        torch.cuda.synchronize()
        for r in range(a_.rounds):
            for n, f in cands:
                res_tab.setdefault((what, kind, M, N, K), {}).setdefault(n, []).append(fl / timeit(f, a_.iters) / 1e12)
        d = res_tab[(what, kind, M, N, K)]
        print(f"{what:15s} {kind:9s} M={M} N={N:6d} K={K:6d}: " + "  ".join(f"{n} {min(v):6.0f}-{max(v):6.0f}" for n, v in d.items()), flush=True)
    names = list(next(iter(res_tab.values())).keys())
    print("sum over the shapes (us), from the median TF/s per shape:")
    for n in names:
        tot = fl_tot = 0.0
        for (what, kind, M, N, K), d in res_tab.items():
            v = sorted(d[n])
            tot += 2.0 * M * N * K / (v[len(v) // 2] * 1e12)
            fl_tot += 2.0 * M * N * K
        print(f"  {n:12s} {tot * 1e6:9.1f} us   {fl_tot / tot / 1e12:7.0f} TF/s", flush=True)
