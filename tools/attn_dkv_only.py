#!/usr/bin/env python3
"""Dev tool (GPU box): time the d = 128 dK/dV kernel alone (B 4, S 2048, 32 / 8 heads, causal) -- used with diagnostic builds of the library
(parts of the kernel's work removed: results are garbage, the TIME shows which pipe binds)."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import lxt_amd.ops as ops  # noqa: E402

B, S, Hq, Hkv, d = 4, 2048, 32, 8, 128
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda").bfloat16()   # noqa: E731
q, k, v, Go = rn(B * S, Hq * d), rn(B * S, Hkv * d), rn(B * S, Hkv * d), rn(B * S, Hq * d)
o, lse = torch.empty_like(q), torch.empty(B, Hq, S, device="cuda")
sc = d ** -0.5
ops.attn_fwd(q, k, v, None, o, lse, B, S, Hq, Hkv, d, sc, True, 0)
Gho, D = torch.empty_like(q), torch.empty(B, Hq, S, device="cuda")
ops.attn_bwd_prep(Go, o, Gho, D, B, S, Hq, d, 0.0, 0.5)
dk, dv = torch.empty_like(q), torch.empty_like(q)
f = lambda: ops.attn_bwd_dkv(q, k, v, None, Gho, None, lse, D, dk, dv, B, S, Hq, Hkv, d, sc, 0.0, 0.0, True, 0)  # noqa: E731
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record()
torch.cuda.synchronize()
print(f"dK/dV d=128 B4 S2048 32/8 causal: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
