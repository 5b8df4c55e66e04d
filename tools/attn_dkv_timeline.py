"""dev (GPU box): segment totals of the d = 128 dK/dV kernel built with -DA32_TIMELINE (a tools/ab library put in place of the product library):
per-wave shader-clock totals of the loop segments, written over the first words of the wave's first dK row."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from lxt_amd import ops

B, S, Hq, Hkv, d = 4, 2048, 32, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").bfloat16()
q, k, v, Go = r(B * S, Hq * d), r(B * S, Hkv * d), r(B * S, Hkv * d), r(B * S, Hq * d)
o, lse = torch.empty_like(q), torch.empty(B, Hq, S, device="cuda")
ops.attn_fwd(q, k, v, None, o, lse, B, S, Hq, Hkv, d, d ** -0.5, True, 0)
Gho, D = torch.empty_like(q), torch.empty(B, Hq, S, device="cuda")
ops.attn_bwd_prep(Go, o, Gho, D, B, S, Hq, d, 0.0, 0.5)
dk, dv = torch.empty_like(q), torch.empty_like(q)
f = lambda: ops.attn_bwd_dkv(q, k, v, None, Gho, None, lse, D, dk, dv, B, S, Hq, Hkv, d, d ** -0.5, 0.0, 0.0, True, 0)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record()
torch.cuda.synchronize()
print(f"dK/dV (timeline build) {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch")
w = dk.view(torch.int32).view(B, S, Hq, d // 2)[:, ::32, :, :8].long().cpu()      # [B, S/32 (waves along the keys), Hq, 8]
names = ["loop+stage", "S/dP", "elementwise", "dV/dK", "vmcnt", "barrier", "total", "tiles"]
for lo, hi_ in ((0, 8), (0, 64), (24, 32), (56, 64)):
    x = w[:, lo:hi_].reshape(-1, 8).double()
    t = x[:, 6].mean()
    print(f"waves of key blocks {lo}..{hi_ - 1}: tiles {x[:, 7].mean():.1f}  total {t:.0f} cyc  per 64-query tile {t / x[:, 7].mean():.0f}  |  " +
          "  ".join(f"{n} {100 * x[:, i].mean() / t:.1f}% ({x[:, i].mean() / x[:, 7].mean():.0f}/tile)" for i, n in enumerate(names[:6])))
