"""Gradient-modifier primitives of the efficient (Gradient x Input) AttnLRP formulation, HIP-backed.

Same names / argument meaning as the reference (ref: lxt/efficient/rules.py:19-66):
    identity_rule_implicit(fn, input)   identity rule on an element-wise non-linearity
    divide_gradient(input, factor=2)    uniform rule after a matmul / element-wise product
    stop_gradient(input)                CP-LRP: no relevance through this tensor
Forward values are whatever `fn` computes; the backward runs on the HIP kernels of liblrp_hip.so
(lrp_act_bwd for the known activations, lrp_mul + lrp_eps_scale for an arbitrary fn).
"""
import torch
from torch.autograd import Function

from .. import ops

_KNOWN_ACT = {}


def _act_name(fn):
    """map a callable onto one of the activations the fused kernels implement (or None)"""
    import torch.nn as nn
    import torch.nn.functional as F
    owner = getattr(fn, "__self__", None)          # bound method of an activation module (non_linear_forward keeps
    if isinstance(owner, (nn.SiLU, nn.GELU)) and getattr(fn, "__name__", "") in ("forward", "original_forward"):   # the original forward)
        fn = owner
    if fn in (F.silu,) or isinstance(fn, nn.SiLU):
        return "silu"
    if isinstance(fn, nn.GELU):
        return "gelu_tanh" if getattr(fn, "approximate", "none") == "tanh" else "gelu"
    if fn is F.gelu:
        return "gelu"
    name = type(fn).__name__
    return {"SiLUActivation": "silu", "GELUTanh": "gelu_tanh", "PytorchGELUTanh": "gelu_tanh", "GELUActivation": "gelu",
            "NewGELUActivation": "gelu_tanh"}.get(name)


class identity_rule_implicit_fn(Function):
    """forward y = fn(x); backward G_in = G_out * y / (x + eps)   (ref: lxt/efficient/rules.py:88-100)"""

    @staticmethod
    def forward(ctx, fn, input, epsilon=1e-10):
        act = _act_name(fn)
        if act is not None and input.is_cuda:
            output = ops.act_fwd(input, act)
        else:
            output = fn(input)
            act = None
        ctx.act, ctx.epsilon = act, epsilon
        if input.requires_grad:
            ctx.save_for_backward(input, output)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        x, y = ctx.saved_tensors
        if ctx.act is not None:
            return None, ops.act_bwd(grad_out, x, ctx.act, ctx.epsilon), None
        r = ops.mul(grad_out, y)
        return None, ops.eps_scale(r, x, 1.0, ctx.epsilon, relevance=True), None


class divide_gradient_fn(Function):
    """identity forward; backward G / factor   (ref: lxt/efficient/rules.py:119-127)"""

    @staticmethod
    def forward(ctx, input, factor=2):
        ctx.factor = factor
        return input

    @staticmethod
    def backward(ctx, grad_out):
        g = grad_out.contiguous()
        return ops.eps_scale(g, g, float(ctx.factor), 0.0), None


def identity_rule_implicit(fn, input):
    return identity_rule_implicit_fn.apply(fn, input)


def divide_gradient(input, factor=2):
    return divide_gradient_fn.apply(input, factor)


def stop_gradient(input):
    return input.detach()
