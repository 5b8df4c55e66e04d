"""CPU: lxt_amd.explicit.core.Composite -- the parts of ref lxt/explicit/core.py that need no kernel: canonizers are applied before the rules and
removed with them (:63-72, :352-356), a canonizer CLASS is refused (:35-37), module rules by type and by NAME (:94-106), function rules through
torch.fx on a traceable module (:155-229: `call_function` targets re-targeted, never inside a module that already carries a rule), remove()."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from lxt_amd.explicit.core import Composite
from lxt_amd.explicit import rules


class _Canon:
    """duck-typed canonizer (zennit / lxt protocol): apply(model[, verbose]) -> instances with .remove()"""

    def __init__(self, takes_verbose=True):
        self.takes_verbose, self.removed = takes_verbose, 0

    def apply(self, model, *args):
        if not self.takes_verbose and args:
            raise TypeError("apply() takes 2 positional arguments")
        self.saw_linear_unwrapped = isinstance(model.fc1, nn.Linear)       # canonizers run BEFORE the rules are attached
        model.tag = getattr(model, "tag", 0) + 1
        return [self]

    def remove(self):
        self.removed += 1


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1, self.fc2, self.act = nn.Linear(8, 8), nn.Linear(8, 4), nn.ReLU()

    def forward(self, x, scale=2.0):
        h = torch.add(self.fc1(x), x)
        h = self.act(h)
        p = F.softmax(torch.matmul(h, h.transpose(-1, -2)), dim=-1)
        return self.fc2(torch.matmul(p, h)) * scale


def test_canonizers_are_applied_first_and_removed():
    m = _Net()
    c1, c2 = _Canon(True), _Canon(False)
    comp = Composite({nn.ReLU: rules.IdentityRule}, canonizers=[c1, c2])
    comp.register(m)
    assert m.tag == 2 and c1.saw_linear_unwrapped and isinstance(m.act, rules.IdentityRule)
    assert all(not p.requires_grad for p in m.parameters())
    comp.remove()
    assert (c1.removed, c2.removed) == (1, 1) and isinstance(m.act, nn.ReLU) and comp.canonizer_instances == []
    with pytest.raises(ValueError):
        Composite({}, canonizers=[_Canon])          # the class instead of an instance


def test_module_rules_by_type_and_by_name():
    m = _Net()
    comp = Composite({"fc2": rules.StopRelevanceRule, nn.ReLU: rules.IdentityRule})
    comp.register(m)
    assert isinstance(m.fc2, rules.StopRelevanceRule) and isinstance(m.act, rules.IdentityRule) and isinstance(m.fc1, nn.Linear)
    comp.remove()
    assert isinstance(m.fc2, nn.Linear)
    with pytest.raises(ValueError):
        Composite({3: rules.IdentityRule}).register(m)


def test_function_rules_rewrite_the_traced_graph():
    calls = {"add": 0, "matmul": 0, "softmax": 0}

    def my_add(a, b):
        calls["add"] += 1
        return torch.add(a, b)

    def my_matmul(a, b):
        calls["matmul"] += 1
        return torch.matmul(a, b)

    def my_softmax(x, dim=-1, **kw):                                    # torch.fx records F.softmax with its private keywords too
        calls["softmax"] += 1
        return F.softmax(x, dim=dim)

    m = _Net().eval()
    x = torch.randn(3, 8)
    ref = m(x)
    comp = Composite({nn.ReLU: rules.IdentityRule, torch.add: my_add, torch.matmul: my_matmul, F.softmax: my_softmax})
    traced = comp.register(m, dummy_inputs={"x": x}, verbose=False)
    assert isinstance(traced, torch.fx.GraphModule)
    out = traced(x)
    assert torch.allclose(out, ref, atol=1e-6) and calls == {"add": 1, "matmul": 2, "softmax": 1}
    assert "replaced" in comp.function_summary["Root"].values()            # functions called in the root forward (ref: "Root")
    with pytest.raises(ValueError):
        Composite({torch.add: my_add}).register(_Net())                 # function rules need dummy_inputs
    comp.remove()
    assert isinstance(m.act, nn.ReLU)
