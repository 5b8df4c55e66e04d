#!/usr/bin/env python3
"""Golden fixture for lxt.explicit.modules.MultiheadAttention_CP, generated FROM THE REAL REFERENCE
(build container only):   python tests/golden/make_golden_mha.py
torch.nn.MultiheadAttention(256, 4, batch_first) built from a seed (weights are not stored), swapped with the reference's
INIT_MODULE_MAPPING, EpsilonRule attached to v_proj / out_proj as in the reference's own test (tests/test_modules.py:42-121),
relevance seeded with the output itself.  Cases: additive float attn_mask, boolean key_padding_mask, no mask."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import lxt.explicit.modules as lm      # noqa: E402
import lxt.explicit.rules as rules      # noqa: E402


def build(seed=21):
    torch.manual_seed(seed)
    return nn.MultiheadAttention(256, 4, batch_first=True).eval()


def inputs(seed=22):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, 20, 256, generator=g)
    attn_mask = torch.randn(20, 20, generator=g)
    kpm = torch.zeros(2, 20, dtype=torch.bool)
    kpm[0, 15:] = True
    kpm[1, 11:] = True
    return x, attn_mask, kpm


def run(dtype, x0, kw):
    gt = build().to(dtype)
    layer = lm.INIT_MODULE_MAPPING[lm.MultiheadAttention_CP](gt, lm.MultiheadAttention_CP)
    kw = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in kw.items()}
    y_gt, a_gt = gt(x0.to(dtype), x0.to(dtype), x0.to(dtype), **kw)
    layer.v_proj = rules.EpsilonRule(layer.v_proj)
    layer.out_proj = rules.EpsilonRule(layer.out_proj)
    x = x0.clone().to(dtype).requires_grad_()
    y, attn = layer(x, x, x, **kw)
    assert torch.allclose(y_gt, y, atol=1e-5) and torch.allclose(a_gt, attn, atol=1e-5)
    y.backward(y)
    return y.detach(), attn.detach(), x.grad


if __name__ == "__main__":
    out = dict(wsum=float(sum(p.detach().double().abs().sum() for p in build().parameters())))
    for name in ("mask", "kpm", "none"):
        # R/(z+eps) has a pole at z = -eps: keep instances on which the reference's own fp32 and fp64 runs agree (SURVEY finding 3)
        for seed in range(22, 80):
            x0, attn_mask, kpm = inputs(seed)
            kw = dict(mask=dict(attn_mask=attn_mask), kpm=dict(key_padding_mask=kpm), none={})[name]
            y, attn, R = run(torch.float32, x0, kw)
            R64 = run(torch.float64, x0, kw)[2]
            gap = float((R.double() - R64).abs().max() / R64.abs().max())
            if gap < 5e-6:
                break
        print(f"  [mha_cp/{name}] input seed {seed}  |y| {float(y.abs().mean()):.4f}  sum R_in {float(R.sum()):.5f}  sum R_out {float(y.sum()):.5f}  "
              f"reference fp32-vs-fp64 {gap:.1e}")
        out.update({f"{name}_x": x0.numpy(), f"{name}_y": y.numpy(), f"{name}_attn": attn.numpy(), f"{name}_R": R.numpy(),
                    f"{name}_R_fp64": R64.float().numpy(), f"{name}_cond_gap": gap})
        if name == "mask":
            out["attn_mask"] = attn_mask.numpy()
        if name == "kpm":
            out["key_padding_mask"] = kpm.numpy()
    np.savez_compressed(os.path.join(HERE, "mha_cp.npz"), **out)
