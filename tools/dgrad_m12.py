"""Dev tool (GPU): the eps-rule dgrad at M = 1, 2 -- lane-local small-M kernel vs the MFMA weight-streaming kernel (stabiliser fused in both), three
weights rotated (HBM figures), both layer-sized weights."""
import sys, torch
sys.path.insert(0, ".")
import lxt_amd.ops as ops
bf, dev = torch.bfloat16, "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def timed(fn, n=21):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, K) in ((14336, 4096), (4096, 14336)):
    Ws = [(torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(bf) for _ in range(3)]
    for M in (1, 2, 3, 4):
        gg = torch.randn(M, N, generator=g, device=dev).to(bf); z = torch.randn(M, N, generator=g, device=dev).to(bf)
        out = torch.empty(M, K, device=dev, dtype=bf)
        a = timed(lambda i: ops.linear_smallm_dgrad(gg, Ws[i % 3], z=z, eps=1e-6, out=out)) if M <= 2 else float("nan")
        b = timed(lambda i: ops.linear_stream_dgrad(gg, Ws[i % 3], z=z, eps=1e-6, out=out))
        r1 = ops.linear_stream_dgrad(gg, Ws[0], z=z, eps=1e-6).float()
        r0 = ops.linear_smallm_dgrad(gg, Ws[0], z=z, eps=1e-6).float() if M <= 2 else r1
        print(f"W [{N},{K}] M {M}: smallm {a:6.1f} us | stream {b:6.1f} us | max rel diff {float((r1 - r0).abs().max() / r0.abs().max()):.2e}")
