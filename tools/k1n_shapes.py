"""Dev tool (GPU): K1n per shape -- each GEMM of a Llama-3-8B layer that carries a norm / residual epilogue against the stand-alone pair it
replaces (GEMM + add_rmsnorm_fwd / rmsnorm_bwd_add2), M = 8192, HIP events, interleaved repeats.   python tools/k1n_shapes.py [M]"""
import sys
import torch
sys.path.insert(0, ".")
import lxt_amd.ops as ops
import lxt_amd.engine as E

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
RPAD = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # extra elements per row of the residual-stream operands (h, out, Gres)
H, I, NQKV = 4096, 14336, 6144
bf, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda r, c, pad=0, sc=1.0: (torch.randn(r, c + pad, generator=g, device=dev) * sc).to(bf)[:, :c]      # noqa: E731
wpad = lambda rows, cols: E.weight_pitch_pad(cols, 2, rows)      # noqa: E731
Wo, Wd = rn(H, H, sc=H ** -0.5), rn(H, I, E.pitch_pad(I, 2), sc=I ** -0.5)
Wqkv, Wgu = rn(NQKV, H, wpad(NQKV, H), sc=H ** -0.5), rn(2 * I, H, wpad(2 * I, H), sc=H ** -0.5)
o, h, m = rn(M, H), rn(M, H, RPAD), rn(M, I, E.pitch_pad(I, 2))
Aqkv, Agu = rn(M, NQKV), rn(M, 2 * I, E.pitch_pad(2 * I, 2))
ones = torch.ones(H, dtype=bf, device=dev)
out, out2, x = torch.empty(M, H, dtype=bf, device=dev), torch.empty(M, H + RPAD, dtype=bf, device=dev)[:, :H], torch.empty(M, H, dtype=bf, device=dev)
ssq, rstd = torch.empty(H // 64, M, device=dev), torch.rand(M, device=dev) + 0.5
qkv, gu, mm = torch.empty(M, NQKV, dtype=bf, device=dev), torch.empty(M, 2 * I, dtype=bf, device=dev), torch.empty(M, I, dtype=bf, device=dev)
Gres = rn(M, H, RPAD)


def timed(fn, n=8):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = {
    "o-proj fwd  [h1 = h + o Wo^T, rstd]": (lambda: (ops.linear_fwd(o, Wo, out=out), ops.add_rmsnorm_fwd(h, out, ones, 1e-5, hsum_out=out2, y=x, rstd=rstd)),
                                           lambda: (ops.gemm_res_ssq(o, Wo, h, out2, ssq), ops.rms_rstd(ssq, M, H, 1e-5, rstd)),
                                           lambda: ops.linear_fwd(o, Wo, out=out)),
    "down fwd    [h' = h1 + m Wd^T, rstd]": (lambda: (ops.linear_fwd(m, Wd, out=out), ops.add_rmsnorm_fwd(h, out, ones, 1e-5, hsum_out=out2, y=x, rstd=rstd)),
                                            lambda: (ops.gemm_res_ssq(m, Wd, h, out2, ssq), ops.rms_rstd(ssq, M, H, 1e-5, rstd)),
                                            lambda: ops.linear_fwd(m, Wd, out=out)),
    "qkv fwd     [rstd (h Wqkv^T)]": (lambda: ops.linear_fwd(h, Wqkv, out=qkv), lambda: ops.gemm_nt_rs(h, Wqkv, rstd, qkv), None),
    "gate/up fwd [rstd (h1 Wgu^T), gated rule]": (lambda: ops.gemm_gated_fwd(h, Wgu, gu, mm, "silu"), lambda: ops.gemm_gated_fwd_rs(h, Wgu, rstd, gu, mm, "silu"), None),
    "qkv dgrad   [rstd (Aqkv Wqkv) + Gres]": (lambda: (ops.linear_dgrad(Aqkv, Wqkv, out=out), ops.rmsnorm_bwd_add2(Gres, out, ones, rstd, None, None, out2, None, None, 0.0, 0.0, 0.0)),
                                             lambda: ops.gemm_nn_rs_res(Aqkv, Wqkv, rstd, Gres, out2), lambda: ops.linear_dgrad(Aqkv, Wqkv, out=out)),
    "gate/up dgrad [rstd (Agu Wgu) + Gres]": (lambda: (ops.linear_dgrad(Agu, Wgu, out=out), ops.rmsnorm_bwd_add2(Gres, out, ones, rstd, None, None, out2, None, None, 0.0, 0.0, 0.0)),
                                             lambda: ops.gemm_nn_rs_res(Agu, Wgu, rstd, Gres, out2), lambda: ops.linear_dgrad(Agu, Wgu, out=out)),
}
print(f"M = {M}, residual-stream row pad {RPAD}; us per call: stand-alone pair | K1n | (GEMM alone)")
for name, (pair, fused, alone) in cases.items():
    tp, tf, ta = [], [], []
    for _ in range(3):
        tp.append(timed(pair))
        tf.append(timed(fused))
        if alone:
            ta.append(timed(alone))
    print(f"{name:48s} {min(tp):8.1f} | {min(tf):8.1f} | {min(ta) if ta else float('nan'):8.1f}    (fused - pair = {min(tf) - min(tp):+.1f})")
