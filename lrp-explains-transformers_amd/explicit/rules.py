"""HIP-backed mirror of lxt.explicit.rules: nn.Module wrappers that attach an LRP rule to a wrapped
module / callable (ref: lxt/explicit/rules.py).  nn.Linear inside EpsilonRule takes the fused
lrp_gemm_nt + eps-scale path; any other differentiable callable uses its PyTorch VJP with the
stabilised division and the final (*) input on the HIP element-wise kernels."""
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import ops
from . import functional as lf


class WrapModule(nn.Module):
    """base: holds the wrapped module as .module (ref: rules.py:8-16)"""

    def __init__(self, module):
        super().__init__()
        self.module = module


class identity_fn(Function):
    """R_in = R_out (ref: rules.py:54-78)"""

    @staticmethod
    def forward(ctx, fn, input):
        return fn(input)

    @staticmethod
    @lf.conservation_check_wrap
    def backward(ctx, R_out):
        return None, R_out


class stop_relevance_fn(Function):
    """no relevance to the input (ref: rules.py:99-122)"""

    @staticmethod
    def forward(ctx, fn, input):
        return fn(input)

    @staticmethod
    def backward(ctx, R_out):
        return None, None


class epsilon_lrp_fn(Function):
    """generic Gradient x Input with stabiliser: s = R/(out+eps); R_i = in_i * VJP_i(s)
    (ref: rules.py:170-222).  divisor = number of inputs for the uniform variant."""
    uniform = False

    @staticmethod
    def forward(ctx, fn, epsilon, *inputs):
        return _eps_forward(ctx, fn, epsilon, inputs)

    @staticmethod
    @lf.conservation_check_wrap
    def backward(ctx, R_out):
        return _eps_backward(ctx, R_out, uniform=False)


class uniform_epsilon_lrp_fn(Function):
    """epsilon rule followed by the uniform split over the inputs (ref: rules.py:253-282)"""

    @staticmethod
    def forward(ctx, fn, epsilon, *inputs):
        return _eps_forward(ctx, fn, epsilon, inputs)

    @staticmethod
    @lf.conservation_check_wrap
    def backward(ctx, R_out):
        return _eps_backward(ctx, R_out, uniform=True)


def _eps_forward(ctx, fn, epsilon, inputs):
    requires = [bool(t.requires_grad) for t in inputs]
    ctx.requires_grads = requires
    if not any(requires):                       # gradient-checkpointing first pass (ref: rules.py:192-195)
        return fn(*inputs)
    det = tuple(t.detach().requires_grad_() if t.requires_grad else t for t in inputs)
    with torch.enable_grad():
        outputs = fn(*det)
    ctx.epsilon = epsilon
    ctx.save_for_backward(*[t for t, r in zip(det, requires) if r], outputs)
    return outputs.detach()


def _eps_backward(ctx, R_out, uniform):
    inputs, outputs = ctx.saved_tensors[:-1], ctx.saved_tensors[-1]
    c = float(len(inputs)) if uniform else 1.0
    # R/(out+eps)/n  ==  R/(n*out + n*eps)
    s = ops.eps_scale(R_out.contiguous(), outputs.detach().contiguous(), c, c * ctx.epsilon, relevance=True)
    grads = torch.autograd.grad(outputs, inputs, s)
    rel = iter(ops.mul(g.contiguous(), x.detach().contiguous()) for g, x in zip(grads, inputs))
    return (None, None) + tuple(next(rel) if r else None for r in ctx.requires_grads)


class uniform_rule_fn(Function):
    """R/n to each of the n inputs (ref: rules.py:391-418)"""

    @staticmethod
    def forward(ctx, fn, *inputs):
        ctx.n = len(inputs)
        return fn(*inputs)

    @staticmethod
    @lf.conservation_check_wrap
    def backward(ctx, R_out):
        g = R_out.contiguous()
        r = ops.eps_scale(g, g, float(ctx.n), 0.0)
        return (None,) + tuple(r for _ in range(ctx.n))


class IdentityRule(WrapModule):
    def forward(self, input):
        return identity_fn.apply(self.module, input)


class StopRelevanceRule(WrapModule):
    def forward(self, input):
        return stop_relevance_fn.apply(self.module, input)


class EpsilonRule(WrapModule):
    """ref: rules.py:125-148 (epsilon default 1e-8)"""

    def __init__(self, module, epsilon=1e-8):
        super().__init__(module)
        self.epsilon = epsilon

    def forward(self, *inputs):
        m = self.module
        linear_like = isinstance(m, nn.Linear) or type(m).__name__ in ("LinearInProjection", "LinearOutProjection")
        if linear_like and len(inputs) == 1 and inputs[0].is_cuda:
            return lf.linear_epsilon(inputs[0], m.weight, m.bias, self.epsilon)     # fused MFMA path
        return epsilon_lrp_fn.apply(m, self.epsilon, *inputs)


class UniformEpsilonRule(WrapModule):
    """ref: rules.py:227-250 (epsilon default 1e-6)"""

    def __init__(self, module, epsilon=1e-6):
        super().__init__(module)
        self.epsilon = epsilon

    def forward(self, *inputs):
        return uniform_epsilon_lrp_fn.apply(self.module, self.epsilon, *inputs)


class UniformRule(WrapModule):
    def forward(self, *inputs):
        return uniform_rule_fn.apply(self.module, *inputs)


def identity(fn, input):
    return identity_fn.apply(fn, input)


def epsilon_lrp(fn, epsilon, *inputs):
    return epsilon_lrp_fn.apply(fn, epsilon, *inputs)
