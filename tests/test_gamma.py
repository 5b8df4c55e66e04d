"""Gamma rule (zennit semantics, PARITY UNPINNED: zennit is absent from the image and from the reference tree).
CPU: properties of the oracle restatement.  GPU: the HIP-backed rule against that oracle, and the zennit-composite-style
registration on a ViT built from torch.nn modules."""
import types

import pytest
import torch
from torch import nn

from oracle import rules as orules
from tests.util import nmax


def test_gamma_oracle_properties():
    torch.manual_seed(0)
    x, W = torch.randn(6, 24, dtype=torch.float64), torch.randn(10, 24, dtype=torch.float64)
    b, G = torch.randn(10, dtype=torch.float64) * 0.1, torch.randn(6, 10, dtype=torch.float64)
    # gamma = 0 degenerates to gradient x input (up to the 1e-6 stabiliser)
    z, Gin = orules.gamma_linear_gxi(x, W, b, G, 0.0)
    assert torch.allclose(z, x @ W.T + b) and nmax(Gin, G @ W) < 1e-4
    # no bias: relevance is conserved for every gamma
    for gamma in (0.05, 0.25, 100.0):
        z, Gin = orules.gamma_linear_gxi(x, W, None, G, gamma)
        assert abs(float((x * Gin).sum()) - float((G * z).sum())) < 1e-5 * abs(float((G * z).sum()))
    # gamma -> infinity on non-negative inputs is the z+ rule: only positive weights carry relevance
    xpos = x.abs()
    z, Gin = orules.gamma_linear_gxi(xpos, W, None, G, 1e9)
    Wp = W.clamp(min=0)
    sel = z > 0
    ref = xpos * (((G * z) / (xpos @ Wp.T)).masked_fill(~sel, 0.0) @ Wp) / xpos
    # rows of z < 0 go through the negative branch (W-): compare only where every output of the row is positive
    rows = sel.all(1)
    if rows.any():
        assert nmax(Gin[rows], ref[rows]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-4), (torch.bfloat16, 1e-1)])
@pytest.mark.parametrize("gamma", [0.0, 0.25, 100.0])
def test_gamma_linear_vs_oracle(dtype, tol, gamma):
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from lxt_amd.efficient.gamma import GammaLinearFn
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 50, 192, generator=g).to(dtype)
    W = (torch.randn(160, 192, generator=g) * 192 ** -0.5).to(dtype)
    b = (torch.randn(160, generator=g) * 0.1).to(dtype)
    Go = torch.randn(3, 50, 160, generator=g).to(dtype)
    xc = x.cuda().requires_grad_()
    z = GammaLinearFn.apply(xc, W.cuda(), b.cuda(), gamma, 1e-6, {})
    z.backward(Go.cuda())
    zr, Gr = orules.gamma_linear_gxi(x.double().reshape(-1, 192), W.double(), b.double(), Go.double().reshape(-1, 160), gamma)
    assert nmax(z.reshape(-1, 160), zr) < (1e-5 if dtype == torch.float32 else 2e-2)
    # relevance form (x * G) is the conditioned quantity: G itself divides by x
    R, Rr = (xc * xc.grad).reshape(-1, 192), x.double().reshape(-1, 192) * Gr
    # (gamma = 0: z+ = x+ W + x- W is formed from two GEMMs that cancel -- R / z+ is ill-conditioned where |z| is small)
    print(f"[gamma {gamma} {dtype}] relevance vs oracle {nmax(R, Rr):.2e}")
    assert nmax(R, Rr) < tol, (gamma, nmax(R, Rr))


@pytest.mark.gpu
def test_gamma_composite_on_mini_vit():
    """register / remove on Linear + patch-embedding Conv2d of a ViT (one subprocess: class-level patches are global)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import os, subprocess, sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = r'''
import sys, types, torch
sys.path.insert(0, %r)
from torch import nn
from lxt_amd.efficient import monkey_patch, adopt
from lxt_amd.efficient.models.vit_torch import cp_LRP
from lxt_amd.efficient.gamma import GammaComposite
from tests.golden.hf_models import build_mini_vit
monkey_patch(types.ModuleType("mini_vit"), cp_LRP)
model = adopt(build_mini_vit().cuda())          # plain torch.nn model: its Linear / LayerNorm / Conv2d join the HIP path
x0 = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(32)).cuda()
def explain():
    x = x0.clone().requires_grad_()
    y = model(x); idx = y.argmax(-1)
    y[torch.arange(2, device="cuda"), idx].sum().backward()
    return y.detach(), (x * x.grad).detach()
y_plain, R_plain = explain()
comp = GammaComposite([(nn.Conv2d, 0.0), (nn.Linear, 0.0)])
comp.register(model)
y0, R0 = explain()
comp.remove()
assert torch.equal(y0, y_plain)
e0 = float((R0 - R_plain).abs().max() / R_plain.abs().max())
comp = GammaComposite([(nn.Conv2d, 0.25), (nn.Linear, 0.05)])
comp.register(model)
y1, R1 = explain()
comp.remove()
y2, R2 = explain()
assert torch.equal(y1, y_plain) and torch.isfinite(R1).all()
assert torch.equal(R2, R_plain), "remove() must restore the un-ruled model"
d1 = float((R1 - R_plain).abs().max() / R_plain.abs().max())
print("gamma=0 vs plain %%.2e ; gamma=(0.25, 0.05) changes the heat-map by %%.2e" %% (e0, d1))
assert e0 < 1e-2 and d1 > 1e-2      # gamma = 0 == gradient x input up to the 1e-6 stabilisers of 14 ruled layers
''' % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    print(r.stdout[-400:])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
