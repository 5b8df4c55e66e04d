#!/usr/bin/env python3
"""Dev probe (GPU box): does the ROW PITCH of the operands explain the long-K GEMM deficit?  Times lrp_gemm_nt (product library) and
torch.matmul (hipBLASLt) at M = 8192, N = 4096 on K-slices of operands with different pitches (elements)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from gemm_ab import load, timeit  # noqa: E402

fn = load(sys.argv[1] if len(sys.argv) > 1 else "lrp-explains-transformers_amd/liblrp_hip.so")
M, N = 8192, 4096
st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731


def run(N, K, pa, pb, pc, iters=10, rounds=3, nn=False):
    """pa / pb / pc: padding (elements) of the A / B / C row pitch; nn: B is [K, N] (the dgrad form on the stored weight)"""
    A = torch.randn(M, K + pa, device="cuda").bfloat16()
    B = (torch.randn(K, N + pb, device="cuda") if nn else torch.randn(N, K + pb, device="cuda")).mul_(K ** -0.5).bfloat16()
    a, b = A[:, :K], (B[:, :N] if nn else B[:, :K])
    OUT = torch.empty(M, N + pc, device="cuda", dtype=torch.bfloat16)
    out = OUT[:, :N]

    def ours():
        if nn:
            rc = fnn(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, a.stride(0), b.stride(0), out.stride(0), 1, 1, st())
        else:
            rc = fn(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, M, N, K, a.stride(0), b.stride(0), out.stride(0), 1, 0, 0, 0, 1, 1, st())
        assert rc == 0, rc

    def blt():
        torch.matmul(a, b if nn else b.T, out=out)
    res = {"ours": [], "hipblaslt": []}
    for _ in range(2):
        ours(); blt()
    for _ in range(rounds):
        for name, f in (("ours", ours), ("hipblaslt", blt)):
            res[name].append(2.0 * M * N * K / timeit(f, iters) * 1e-12)
    ref = a[:64].float() @ (b.float() if nn else b.float().T)
    ours()
    err = float((out[:64].float() - ref).abs().max() / ref.abs().max())
    print(f"{'NN' if nn else 'NT'} N={N:6d} K={K:6d} pad A/B/C = {pa}/{pb}/{pc}: ours {min(res['ours']):.0f}-{max(res['ours']):.0f}  hipblaslt "
          f"{min(res['hipblaslt']):.0f}-{max(res['hipblaslt']):.0f} TF/s  (err {err:.1e})", flush=True)


import ctypes  # noqa: E402
_lib = ctypes.CDLL(os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "lrp-explains-transformers_amd/liblrp_hip.so"))
fnn = _lib.lrp_gemm_nn
fnn.restype = ctypes.c_int
fnn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
NT_CASES = [(N, K, pads) for N, K in [(4096, 4096), (6144, 4096), (28672, 4096), (4096, 6144)] for pads in [(0, 0, 0), (64, 64, 0), (0, 0, 64), (64, 64, 64)]] + \
    [(N, K, pads) for N, K in [(4096, 14336), (4096, 28672)] for pads in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0)]]
if '--nt' in sys.argv:
    for N, K, pads in NT_CASES:
        run(N, K, *pads)
# dgrad forms on the stored weights: down bwd (Wd [4096, 14336]), gate/up bwd (Wgu [28672, 4096]), qkv bwd, o bwd
for N, K in [(14336, 4096), (4096, 28672), (4096, 6144), (4096, 4096)]:
    for pads in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 64)]:
        run(N, K, *pads, nn=True)
