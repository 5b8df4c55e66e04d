#!/usr/bin/env python3
"""dev (GPU box): row pitch of the stashed gate/up output gu [M, 2 I] (written by the fused gate/up forward, read by the fused down dgrad's epilogue),
of x / Adn (A operands, K = 4096) and of the C outputs: which pitches off the 4-KiB grid pay?  Llama-3-8B layer shapes, M = 8192, interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import lxt_amd.ops as ops

M, H, I = 8192, 4096, 14336
rn = lambda r, c, pad=0: (torch.randn(r, c + pad, device="cuda") * 0.05).bfloat16()[:, :c]      # noqa: E731
def t(f, n=8):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
Wgu, Wd = rn(2 * I, H, 128), rn(H, I, 64)
res = {}
for rnd in range(3):
    for gpad in (0, 64, 128):
        for apad in (0, 64):
            x, Adn = rn(M, H, apad), rn(M, H, apad)
            gu, m, Agu = rn(M, 2 * I, gpad), rn(M, I, 64), rn(M, 2 * I, 64)
            f = t(lambda: ops.gemm_gated_fwd(x, Wgu, gu, m, "silu"))
            b = t(lambda: ops.gemm_gated_bwd(Adn, Wd, gu, Agu, 1e-10, 0.0, "silu"))
            res.setdefault((gpad, apad), []).append((f, b))
            del x, Adn, gu, m, Agu
for k, v in res.items():
    print(f"gu pad {k[0]:3d}  x/Adn pad {k[1]:2d}:  gated fwd {min(a for a, _ in v):7.1f}-{max(a for a, _ in v):7.1f} us   gated bwd {min(b for _, b in v):7.1f}-{max(b for _, b in v):7.1f} us", flush=True)
# plain GEMMs: C pitch and A pitch
Wqkv, Wo = rn(6144, H, 128), rn(H, H)
res = {}
for rnd in range(3):
    for cpad in (0, 64):
        for apad in (0, 64):
            x = rn(M, H, apad)
            qkv, o = rn(M, 6144, cpad), rn(M, H, cpad)
            a = t(lambda: ops.gemm_nt_2d(x, Wqkv, qkv))
            b = t(lambda: ops.gemm_nt_2d(x, Wo, o))
            c = t(lambda: ops.gemm_nn_2d(rn(M, 6144, apad) if False else qkv, Wqkv, o))       # qkv dgrad: A = [M, 6144] with pitch 6144 + cpad
            res.setdefault((cpad, apad), []).append((a, b, c))
for k, v in res.items():
    print(f"C pad {k[0]:2d} (also the qkv-dgrad A pitch)  x pad {k[1]:2d}:  qkv fwd {min(a for a, _, _ in v):6.1f}  o fwd {min(b for _, b, _ in v):6.1f}  qkv dgrad {min(c for _, _, c in v):6.1f} us", flush=True)
