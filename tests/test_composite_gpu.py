"""GPU: Composite FUNCTION rules on the device (ref lxt/explicit/core.py:155-227: the traced graph's torch.matmul / softmax / add calls re-targeted
to lf.matmul / lf.softmax / lf.add2) against the reference's own outputs for those rules (tests/golden/rules.npz: mm_*, sm_*, add_*, captured
from the imported lxt.explicit.functional with consistent relevance R_out = z (*) g).  SURVEY.md 8f-4; closes VERDICT r5 missing item 5: the fx
path had only carried plain torch callables on the CPU."""
import operator

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.util import nmax, load, t

pytestmark = pytest.mark.gpu


class _MatMul(nn.Module):
    def forward(self, a, b):
        return torch.matmul(a, b)


class _Soft(nn.Module):
    def forward(self, x):
        return F.softmax(x, dim=-1)


class _Add(nn.Module):
    def forward(self, a, b):
        return torch.add(a, b)


class _Block(nn.Module):
    """one traced graph carrying all three function rules and a module rule: y = W (softmax(a b) b^T + a)"""

    def __init__(self):
        super().__init__()
        self.proj = nn.Linear(32, 32, bias=False)

    def forward(self, a, b):
        p = F.softmax(torch.matmul(a, b), dim=-1)
        return self.proj(torch.add(torch.matmul(p, b.transpose(-1, -2)), a))


def _rules():
    import lxt_amd.explicit.functional as lf
    import lxt_amd.explicit.rules as rules
    return lf, rules, {torch.matmul: lf.matmul, F.softmax: lf.softmax, torch.add: lf.add2, operator.add: lf.add2}


def test_function_rules_on_the_device_match_the_reference_fixtures():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from lxt_amd.explicit.core import Composite
    lf, rules, fmap = _rules()
    fx = load("rules.npz")
    dev = "cuda"
    # ---- lf.matmul through the traced graph (Prop. 3.3; ref lxt/explicit/functional.py:367-408, wrapper eps 1e-8)
    a, b, g = (t(fx[k]).to(dev) for k in ("mm_a", "mm_b", "mm_g"))
    comp = Composite(dict(fmap))
    traced = comp.register(_MatMul(), dummy_inputs={"a": a, "b": b})
    assert isinstance(traced, torch.fx.GraphModule) and comp.function_summary["Root"][torch.matmul] == "replaced"
    assert any(n.op == "call_function" and n.target is lf.matmul for n in traced.graph.nodes)
    a_, b_ = a.clone().requires_grad_(), b.clone().requires_grad_()
    o = traced(a_, b_)
    assert nmax(o, torch.matmul(a.double(), b.double())) < 1e-5
    Ra, Rb = torch.autograd.grad(o, (a_, b_), o.detach() * g)
    assert nmax(Ra, fx["mm_Ra"]) < 2e-5 and nmax(Rb, fx["mm_Rb"]) < 2e-5
    # ---- lf.softmax (Prop. 3.1 incl. -inf entries; ref :276-322)
    x, g = t(fx["sm_x"]).to(dev), t(fx["sm_g"]).to(dev)
    traced = Composite(dict(fmap)).register(_Soft(), dummy_inputs={"x": x})
    x_ = x.clone().requires_grad_()
    p = traced(x_)
    assert nmax(p, fx["sm_p"]) < 1e-6
    Rx, = torch.autograd.grad(p, x_, p.detach() * g)
    assert nmax(Rx, fx["sm_Rx"]) < 2e-5
    # ---- lf.add2 (ref :412-459, wrapper eps 1e-8)
    a, b, g = (t(fx[k]).to(dev) for k in ("add_a", "add_b", "add_g"))
    traced = Composite(dict(fmap)).register(_Add(), dummy_inputs={"a": a, "b": b})
    a_, b_ = a.clone().requires_grad_(), b.clone().requires_grad_()
    s = traced(a_, b_)
    Ra, Rb = torch.autograd.grad(s, (a_, b_), s.detach() * g)
    assert nmax(Ra, fx["add_Ra"]) < 2e-5 and nmax(Rb, fx["add_Rb"]) < 2e-5


def test_traced_block_equals_the_hand_composed_rules():
    """module rule + three function rules in ONE traced graph: the same relevance as calling the lf.* functions by hand (what the reference's
    model files do), and relevance is conserved through the function rules"""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from lxt_amd.explicit.core import Composite
    lf, rules, fmap = _rules()
    gen = torch.Generator().manual_seed(5)
    a = torch.randn(2, 3, 10, 32, generator=gen).cuda()
    b = (torch.randn(2, 3, 32, 10, generator=gen) * 0.3).cuda()
    net = _Block().cuda()
    W = net.proj.weight.detach().clone()
    comp = Composite({nn.Linear: rules.EpsilonRule, **fmap})
    traced = comp.register(net, dummy_inputs={"a": a, "b": b})
    assert isinstance(net.proj, rules.EpsilonRule)
    kinds = {n.target for n in traced.graph.nodes if n.op == "call_function"}
    assert lf.matmul in kinds and lf.softmax in kinds and lf.add2 in kinds and torch.matmul not in kinds
    a1, b1 = a.clone().requires_grad_(), b.clone().requires_grad_()
    y = traced(a1, b1)
    seed = y.detach().clone()
    Ra, Rb = torch.autograd.grad(y, (a1, b1), seed)
    a2, b2 = a.clone().requires_grad_(), b.clone().requires_grad_()
    p = lf.softmax(lf.matmul(a2, b2), dim=-1)
    y2 = rules.EpsilonRule(nn.Linear(32, 32, bias=False).cuda().requires_grad_(False))
    y2.module.weight.copy_(W)
    out2 = y2(lf.add2(lf.matmul(p, b2.transpose(-1, -2)), a2))
    Ra2, Rb2 = torch.autograd.grad(out2, (a2, b2), seed)
    assert nmax(y, out2) < 1e-6 and nmax(Ra, Ra2) < 1e-6 and nmax(Rb, Rb2) < 1e-6
    comp.remove()
    assert isinstance(net.proj, nn.Linear)
