#!/usr/bin/env python3
"""GPU box: row N1 (Linear eps-rule in its HBM-bound regime) per variant, on the two layer-sized weights in the ENGINE's layout (pitch off the 4-KiB
grid), three weights rotated: forward stream / skinny; dgrad = eps_scale + stream | stream with the stabiliser inside | eps_scale + skinny NN;
eps_scale alone.  us per launch; pair fraction of 8 TB/s for the best forward + best dgrad.   Usage: n1_probe.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import lxt_amd.engine as E_  # noqa: E402
import lxt_amd.ops as ops  # noqa: E402


def timed(fn, n=21):
    for i in range(3):
        fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [16, 32, 48, 64, 96, 128, 160]
    g = torch.Generator(device="cuda").manual_seed(0)
    bf = torch.bfloat16
    for (N, K, pad) in ((14336, 4096, E_.weight_pitch_pad(4096, 2, 14336)), (4096, 14336, E_.pitch_pad(14336, 2))):
        Ws = [(torch.randn(N, K + pad, generator=g, device="cuda") * K ** -0.5).to(bf)[:, :K] for _ in range(3)]
        print(f"# W [{N},{K}] bf16 (pitch {K + pad}) x 3 rotated", flush=True)
        for M in Ms:
            x = torch.randn(M, K, generator=g, device="cuda").to(bf)
            gg = torch.randn(M, N, generator=g, device="cuda").to(bf)
            z = ops.linear_fwd(x, Ws[0])
            c = torch.empty(M, K, device="cuda", dtype=bf)
            s = torch.empty_like(gg)
            r = {}
            for rep in range(2):
                ops.STREAM_FWD = True
                r["f_stream"] = timed(lambda i: ops.linear_fwd(x, Ws[i % 3], out=z)) if ops.linear_stream_ok(x, Ws[0]) else float("nan")
                r["f_skinny"] = timed(lambda i: ops.gemm_skinny(x, Ws[i % 3], z, nn=False))
                r["eps"] = timed(lambda i: ops.eps_scale(gg, z, 1.0, 1e-6, out=s))
                if M <= 64:
                    r["d_stream"] = timed(lambda i: ops.linear_stream_dgrad(ops.eps_scale(gg, z, 1.0, 1e-6, out=s), Ws[i % 3], out=c))
                    r["d_stream_only"] = timed(lambda i: ops.linear_stream_dgrad(s, Ws[i % 3], out=c))
                if M <= 32:
                    r["d_fused"] = timed(lambda i: ops.linear_stream_dgrad(gg, Ws[i % 3], z=z, eps=1e-6, out=c))
                r["d_skinny"] = timed(lambda i: ops.gemm_skinny(ops.eps_scale(gg, z, 1.0, 1e-6, out=s), Ws[i % 3], c, nn=True))
                r["d_skinny_only"] = timed(lambda i: ops.gemm_skinny(s, Ws[i % 3], c, nn=True))
            bfw, bbw = 2 * (N * K + M * K + M * N), 2 * (N * K + M * K + 2 * M * N)
            fbest = min(v for k, v in r.items() if k.startswith("f_") and v == v)
            dbest = min(v for k, v in r.items() if k in ("d_stream", "d_fused", "d_skinny"))
            print(f"M={M:4d} " + " ".join(f"{k} {v:6.2f}" for k, v in r.items()) + f" | pair {(bfw + bbw) / (fbest + dbest) / 1e6 / 8.0:5.3f}", flush=True)
        del Ws


if __name__ == "__main__":
    main()
