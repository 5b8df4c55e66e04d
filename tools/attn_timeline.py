"""dev: segment totals of the dQ kernel built with -DA32_TIMELINE (tools/ab/dev_timeline.so put in place of the product library)."""
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
dq = torch.empty_like(q)
for _ in range(3):
    ops.attn_bwd_dq(q, k, v, None, Gho, lse, D, dq, B, S, Hq, Hkv, d, d ** -0.5, 0.0, 0.0, True, 0)
torch.cuda.synchronize()
w = dq.view(torch.int32).view(B, S, Hq, d // 2)[:, ::32, :, :8].long().cpu()      # [B, S/32, Hq, 8]
names = ["loop+stage", "S/dP", "elementwise", "dQ", "vmcnt", "barrier", "total", "tiles"]
for lo, hi_ in ((0, 64), (32, 64), (60, 64), (0, 8)):
    x = w[:, lo:hi_].reshape(-1, 8).double()
    t = x[:, 6].mean()
    print(f"waves with query block {lo}..{hi_ - 1}: tiles {x[:, 7].mean():.1f}  total {t:.0f} cyc  per tile {t / x[:, 7].mean():.0f}  |  " +
          "  ".join(f"{n} {100 * x[:, i].mean() / t:.1f}%" for i, n in enumerate(names[:6])))
