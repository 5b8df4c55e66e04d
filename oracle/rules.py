"""Rule-level oracle: closed-form CPU restatements of the reference's LRP rules.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function takes plain tensors and
returns (forward_output, input_relevance...) computed with explicit formulas -- no
autograd.Function, so the arithmetic is visible.  Citations are to /root/reference.

Conventions: ``R_out`` is the relevance arriving at the op's output, ``R_*`` the
relevance assigned to its inputs.  ``stabilize(z, eps) = z + eps`` is UNSIGNED, exactly
as the reference does (lxt/explicit/functional.py:266-273).
"""
import torch
import torch.nn.functional as F


def stabilize(z, eps):
    # lxt/explicit/functional.py:266-273 (_stabilize): x + eps, no sign handling
    return z + eps


# ----------------------------------------------------------------------------- explicit rules
def linear_epsilon(x, weight, bias, R_out, eps=1e-6):
    """lxt/explicit/functional.py:345-364 (linear_epsilon_fn); generic twin
    lxt/explicit/rules.py:188-222 (epsilon_lrp_fn, eps default 1e-8).
    z = x W^T + b ; R_in = x * ((R_out / (z + eps)) W)"""
    z = F.linear(x, weight, bias)
    s = R_out / stabilize(z, eps)
    return z, torch.matmul(s, weight) * x


def matmul(a, b, R_out, eps=1e-8):
    """lxt/explicit/functional.py:385-408 (matmul_fn). s = R/(2 O + eps);
    R_a = (s B^T) * a ; R_b = (A^T s) * b"""
    o = torch.matmul(a, b)
    s = R_out / stabilize(o * 2, eps)
    Ra = torch.matmul(s, b.transpose(-1, -2)) * a
    Rb = torch.matmul(a.transpose(-1, -2), s) * b
    return o, Ra, Rb


def softmax(x, R_out, dim=-1, temperature=1.0):
    """lxt/explicit/functional.py:293-322 (softmax_fn), Prop. 3.1.
    p = softmax(x/T) ; R_in = x' * (R_out - p * sum(R_out)) with -inf -> 0 in x' = x/T."""
    xs = x / temperature
    p = F.softmax(xs, dim=dim)
    xz = torch.where(torch.isneginf(xs), torch.zeros_like(xs), xs)
    return p, xz * (R_out - p * R_out.sum(dim, keepdim=True))


def add2(a, b, R_out, eps=1e-8):
    """lxt/explicit/functional.py:430-459 (add2_tensors_fn). s = R/(a+b+eps)."""
    o = a + b
    s = R_out / stabilize(o, eps)
    return o, s * a, s * b


def mul2(a, b, R_out, a_requires=True, b_requires=True):
    """lxt/explicit/functional.py:517-536 (mul2_fn): uniform split over the inputs that
    require grad; a constant operand gets nothing and the other gets 100 %."""
    n = int(a_requires) + int(b_requires)
    r = R_out / n
    return a * b, (r if a_requires else None), (r if b_requires else None)


def mean(x, R_out, dim=-1, keepdim=True, eps=1e-6):
    """lxt/explicit/functional.py:555-583 (mean_fn)."""
    y = x.mean(dim, keepdim)
    xs = x.sum(dim, keepdim=True)
    r = R_out if keepdim else R_out.unsqueeze(dim)
    return y, x * r / stabilize(xs, eps)


def rms_norm_identity(x, weight, var_eps, R_out):
    """lxt/explicit/functional.py:481-495 (rms_norm_identity_fn): fp32 forward,
    relevance passes through unchanged."""
    in_dtype = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + var_eps)
    return weight * h.to(in_dtype), R_out


def layer_norm(x, weight, bias, var_eps, R_out, eps=1e-6):
    """lxt/explicit/functional.py:606-635 (layer_norm_grad_fn): differentiate
    y=(x-mean)/std.detach()*w+b ; R_in = x * VJP(R_out/(y+eps)).
    VJP of the detached-std layer: u = s*w/std ; grad = u - mean_row(u)."""
    mean_ = x.mean(-1, keepdim=True)
    var = ((x - mean_) ** 2).mean(-1, keepdim=True)
    std = (var + var_eps).sqrt()
    y = (x - mean_) / std
    if weight is not None:
        y = y * weight
    if bias is not None:
        y = y + bias
    s = R_out / stabilize(y, eps)
    u = s / std
    if weight is not None:
        u = u * weight
    g = u - u.mean(-1, keepdim=True)
    return y, g * x


def identity(fn, x, R_out):
    """lxt/explicit/rules.py:68-78 (identity_fn): R_in = R_out."""
    return fn(x), R_out


def uniform(fn, inputs, R_out):
    """lxt/explicit/rules.py:405-418 (uniform_rule_fn): R/n to each input."""
    n = len(inputs)
    return fn(*inputs), tuple(R_out / n for _ in range(n))


def uniform_epsilon_matmul(p, v, R_out, eps=1e-6):
    """lxt/explicit/rules.py:267-282 (uniform_epsilon_lrp_fn) wrapping torch.matmul
    (AttentionValueMatmul, lxt/explicit/models/llama.py:79-81,88):
    s = R/(O+eps)/2 ; R_p = (s V^T) * p ; R_v = (P^T s) * v."""
    o = torch.matmul(p, v)
    s = R_out / stabilize(o, eps) / 2
    return o, torch.matmul(s, v.transpose(-1, -2)) * p, torch.matmul(p.transpose(-1, -2), s) * v


# ----------------------------------------------------------------------------- efficient primitives
def identity_rule_implicit(fn, x, G_out, eps=1e-10):
    """lxt/efficient/rules.py:88-100: forward y=f(x); backward G_in = G_out * y/(x+eps)."""
    y = fn(x)
    return y, G_out * (y / (x + eps))


def divide_gradient(x, G_out, factor=2):
    """lxt/efficient/rules.py:119-127: identity forward, G/factor backward."""
    return x, G_out / factor


# ----------------------------------------------------------------------------- Gamma rule (zennit) -- PARITY UNPINNED
def zennit_stabilize(x, eps):
    """zennit.core.stabilize: x + eps * sign(x), with sign(0) := +1 (signed stabiliser, unlike lxt's own)"""
    return x + ((x == 0.).to(x) + x.sign()) * eps


def gamma_linear_gxi(x, weight, bias, G_out, gamma, eps=1e-6):
    """Generalised Gamma rule for a Linear layer in lxt's gradient x input framework.

    PARITY UNPINNED: the arithmetic lives in the third-party package zennit (setup.py:18 of the reference, un-pinned,
    not vendored, NOT installed here), so this is a restatement of zennit's published rule (zennit/rules.py `Gamma`,
    release 0.5.x: four modified passes + the plain one) composed with the reference's own hook changes
    (lxt/efficient/zennit_patches.py:32-62: relevance = grad_output * output on entry, / stabilize(input, 1e-10) on exit).
    No reference test or fixture touches it.

      pass  input        weight              bias
      0     x+ = max(x,0)  W + g*max(W,0)      b + g*max(b,0)
      1     x- = min(x,0)  W + g*min(W,0)      --
      2     x+             W + g*min(W,0)      b + g*min(b,0)
      3     x-             W + g*max(W,0)      --
      z = x W^T + b ; R = G*z ; s+ = [z>0] R / stab(o0+o1) ; s- = [z<0] R / stab(o2+o3)
      R_in = x+*(s+ Wp) + x-*(s+ Wm) + x+*(s- Wm) + x-*(s- Wp) ;  G_in = R_in / stab(x, 1e-10)
    Returns (z, G_in)."""
    xp, xm = x.clamp(min=0), x.clamp(max=0)
    Wp, Wm = weight + gamma * weight.clamp(min=0), weight + gamma * weight.clamp(max=0)
    bp = bm = None
    if bias is not None:
        bp, bm = bias + gamma * bias.clamp(min=0), bias + gamma * bias.clamp(max=0)
    z = F.linear(x, weight, bias)
    zpos = F.linear(xp, Wp, bp) + F.linear(xm, Wm)
    zneg = F.linear(xp, Wm, bm) + F.linear(xm, Wp)
    R = G_out * z
    sp = (z > 0).to(z) * R / zennit_stabilize(zpos, eps)
    sn = (z < 0).to(z) * R / zennit_stabilize(zneg, eps)
    R_in = xp * (sp @ Wp) + xm * (sp @ Wm) + xp * (sn @ Wm) + xm * (sn @ Wp)
    return z, R_in / zennit_stabilize(x, 1e-10)
