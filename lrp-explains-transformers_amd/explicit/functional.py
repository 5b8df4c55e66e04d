"""HIP-backed mirror of lxt.explicit.functional: every rule is an autograd.Function whose backward
receives RELEVANCE at the output and returns relevance at the inputs (ref: lxt/explicit/functional.py).
Forward and backward both run on liblrp_hip.so; the backward uses the forward's own saved output z
(never a recomputed one -- SURVEY.md finding 5).  Defaults (eps, argument order) are the reference's.
"""
import torch
from torch.autograd import Function

from .. import ops


CONSERVATION_CHECK_FLAG = [False]


def conservation_check_wrap(func):
    """sanity mode of the reference (ref: functional.py:10-37, toggled by lxt.explicit.check.conservation_check): with the flag
    set, a rule's backward hands sum(R_out) spread UNIFORMLY over its inputs instead of the rule's own result, so that
    sum(relevance) at the model input equals the seeded relevance iff every op on the path is LRP-wrapped (bias terms aside)"""
    def wrapped(ctx, *out_relevance):
        inp_relevance = func(ctx, *out_relevance)
        if CONSERVATION_CHECK_FLAG[0]:
            single = not isinstance(inp_relevance, tuple)
            rel = (inp_relevance,) if single else inp_relevance
            total = sum(r.float().sum() for r in out_relevance if r is not None)
            n = sum(r.numel() for r in rel if torch.is_tensor(r))
            mean = total / n
            if torch.isnan(mean).any():
                raise ValueError(f"NaN at {func}")
            rel = tuple(torch.full(r.shape, float(mean), device=r.device, dtype=r.dtype) if torch.is_tensor(r) else None for r in rel)
            inp_relevance = rel[0] if single else rel
        return inp_relevance
    wrapped.__name__ = getattr(func, "__name__", "backward")
    return wrapped


def _stabilize(input, epsilon=1e-6, inplace=False):
    """ref: functional.py:266-273 -- unsigned stabiliser x + eps"""
    return input.add_(epsilon) if inplace else input + epsilon


class linear_epsilon_fn(Function):
    """z = x W^T + b ; R_in = x * ((R_out/(z+eps)) W)      ref: functional.py:345-364"""

    @staticmethod
    def forward(ctx, inputs, weight, bias=None, epsilon=1e-6):
        shp = inputs.shape
        x2 = inputs.reshape(-1, shp[-1]).contiguous()
        w = weight.detach()
        # <= 4 rows: the W-streaming kernels form s = R/(z+eps) on the fly (one launch pair for the whole rule); above that
        # ops.linear_fwd / ops.linear_dgrad (skinny split-K, NN GEMM from the stored weight, or the fp32 path with a cached W^T)
        ctx.small = x2.shape[0] <= 4 and ops.smallm_ok(x2.shape[0], w, x2)
        z = ops.linear_smallm_fwd(x2, w, bias) if ctx.small else ops.linear_fwd(x2, w, bias)
        ctx.save_for_backward(x2, w, z)
        ctx.epsilon, ctx.shp = epsilon, shp
        return z.view(*shp[:-1], weight.shape[0])

    @staticmethod
    @conservation_check_wrap
    def backward(ctx, R_out):
        x2, w, z = ctx.saved_tensors
        if ctx.small:       # s = R/(z+eps), (s W) (*) x in one launch pair
            R_in = ops.linear_smallm_dgrad(R_out.reshape(z.shape).contiguous(), w, z=z, x=x2, eps=ctx.epsilon, relevance_in=True,
                                           relevance_out=True)
            return R_in.view(ctx.shp), None, None, None
        R2 = R_out.reshape(z.shape)
        if 2 < z.shape[0] <= 16 and ctx.epsilon != 0.0 and R2.stride(-1) == 1 and ops.linear_stream_dgrad_ok(R2, w):
            # 3 ... 16 rows, bf16: R / (z + eps) is formed inside the weight-streaming dgrad (one launch for the stabiliser and the contraction)
            R_in = ops.mul(ops.linear_stream_dgrad(R2, w, z=z, eps=ctx.epsilon, relevance_in=True), x2)
            return R_in.view(ctx.shp), None, None, None
        s = ops.eps_scale(R2, z, 1.0, ctx.epsilon, relevance=True)
        R_in = ops.mul(ops.linear_dgrad(s, w), x2)
        return R_in.view(ctx.shp), None, None, None


class matmul_fn(Function):
    """O = A B ; s = R/(2 O + eps) ; R_a = (s B^T)*A ; R_b = (A^T s)*B      ref: functional.py:385-408"""

    @staticmethod
    def forward(ctx, input_a, input_b, inplace=False, epsilon=1e-6):
        a, b = input_a.contiguous(), input_b.contiguous()
        bt = ops.transpose(b)
        o = ops.gemm_nt(a, bt)
        ctx.save_for_backward(a, b, bt, o)
        ctx.epsilon = epsilon
        return o

    @staticmethod
    @conservation_check_wrap
    def backward(ctx, R_out):
        a, b, bt, o = ctx.saved_tensors
        s = ops.eps_scale(R_out, o, 2.0, ctx.epsilon, relevance=True)
        Ra = ops.mul(ops.gemm_nt(s, b), a)                                   # s B^T : contraction over B's columns
        Rb = ops.mul(ops.gemm_nt(ops.transpose(a), ops.transpose(s)), b)     # A^T s
        return Ra, Rb, None, None


class softmax_fn(Function):
    """p = softmax(x/T) ; R_in = (x/T) * (R_out - p * sum R_out), -inf -> 0      ref: functional.py:293-322"""

    @staticmethod
    def forward(ctx, inputs, dim, dtype=None, temperature=1.0, inplace=False):
        if dtype is not None:
            inputs = inputs.to(dtype)
        nd = inputs.dim()
        dim = dim % nd
        xt = inputs.transpose(dim, nd - 1).contiguous() if dim != nd - 1 else inputs.contiguous()
        p = ops.softmax_fwd(xt, 1.0 / temperature)
        ctx.save_for_backward(xt, p)
        ctx.dim, ctx.nd, ctx.inv_t = dim, nd, 1.0 / temperature
        return p.transpose(dim, nd - 1) if dim != nd - 1 else p

    @staticmethod
    @conservation_check_wrap
    def backward(ctx, R_out):
        xt, p = ctx.saved_tensors
        dim, nd = ctx.dim, ctx.nd
        Rt = R_out.transpose(dim, nd - 1).contiguous() if dim != nd - 1 else R_out.contiguous()
        Rx = ops.softmax_rule_bwd(xt, p, Rt.to(xt.dtype), ctx.inv_t)
        return (Rx.transpose(dim, nd - 1) if dim != nd - 1 else Rx), None, None, None, None


class add2_tensors_fn(Function):
    """o = a + b ; s = R/(a+b+eps) ; R_a = s a ; R_b = s b      ref: functional.py:430-459"""

    @staticmethod
    def forward(ctx, input_a, input_b, inplace=False, epsilon=1e-6):
        a, b = torch.broadcast_tensors(input_a, input_b)
        ctx.shapes = (input_a.shape, input_b.shape)
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        ctx.epsilon = epsilon
        ctx.req = (input_a.requires_grad, input_b.requires_grad)
        return a + b

    @staticmethod
    @conservation_check_wrap
    def backward(ctx, R_out):
        a, b = ctx.saved_tensors
        Ra, Rb = ops.add2_rule_bwd(a, b, R_out, ctx.epsilon, need_b=True)
        sa, sb = ctx.shapes
        if Ra.shape != sa:
            Ra = Ra.sum_to_size(sa)
        if Rb.shape != sb:
            Rb = Rb.sum_to_size(sb)
        return Ra, Rb, None, None


class mul2_fn(Function):
    """uniform rule on a product: R/n to each input that requires grad      ref: functional.py:517-536"""

    @staticmethod
    def forward(ctx, input_a, input_b, inplace=False):
        ctx.requires_grads = [i for i, t in enumerate((input_a, input_b)) if isinstance(t, torch.Tensor) and t.requires_grad]
        return input_a * input_b

    @staticmethod
    @conservation_check_wrap
    def backward(ctx, R_out):
        n = len(ctx.requires_grads)
        g = R_out.contiguous()
        r = ops.eps_scale(g, g, float(n), 0.0) if n > 1 else R_out
        return tuple(r if i in ctx.requires_grads else None for i in range(2)) + (None,)


class rms_norm_identity_fn(Function):
    """RMSNorm forward (fp32 statistics), relevance passes through      ref: functional.py:481-495"""

    @staticmethod
    def forward(ctx, hidden_states, weight, variance_epsilon):
        shp = hidden_states.shape
        y, _ = ops.add_rmsnorm_fwd(hidden_states.reshape(-1, shp[-1]).contiguous(), None, weight, float(variance_epsilon))
        return y.view(shp)

    @staticmethod
    @conservation_check_wrap
    def backward(ctx, R_out):
        return R_out, None, None


class layer_norm_grad_fn(Function):
    """y = (x-mean)/std.detach()*w+b ; R_in = x * VJP(R_out/(y+eps))      ref: functional.py:606-635"""

    @staticmethod
    def forward(ctx, x, weight, bias, variance_epsilon, epsilon=1e-6):
        xc = x.contiguous()
        y, _, rstd = ops.layernorm_fwd(xc, weight, bias, float(variance_epsilon))
        ctx.save_for_backward(xc, y, weight, rstd)
        ctx.epsilon = epsilon
        return y

    @staticmethod
    @conservation_check_wrap
    def backward(ctx, R_out):
        x, y, weight, rstd = ctx.saved_tensors
        s = ops.eps_scale(R_out, y, 1.0, ctx.epsilon, relevance=True)      # R/(y+eps)
        g = ops.layernorm_bwd(s, None, weight, rstd, 0.0)                  # VJP of the detached-std layer
        return ops.mul(g, x), None, None, None, None


class mean_fn(Function):
    """epsilon rule for mean: R_in = x * R_out / (sum x + eps)      ref: functional.py:555-583"""

    @staticmethod
    def forward(ctx, x, dim, keepdim, epsilon=1e-6):
        ctx.save_for_backward(x)
        ctx.epsilon, ctx.dim, ctx.keepdim = epsilon, dim, keepdim
        return x.mean(dim, keepdim)

    @staticmethod
    @conservation_check_wrap
    def backward(ctx, R_out):
        (x,) = ctx.saved_tensors
        xs = x.sum(ctx.dim, keepdim=True)
        r = R_out if ctx.keepdim else R_out.unsqueeze(ctx.dim)
        s = ops.eps_scale(r.contiguous(), xs.contiguous(), 1.0, ctx.epsilon, relevance=True)
        return ops.mul(x, s.expand_as(x).contiguous()), None, None, None


class normalize_identity_fn(Function):
    """F.normalize with the identity rule      ref: functional.py:655-665"""

    @staticmethod
    def forward(ctx, input, p, dim, eps):
        return torch.nn.functional.normalize(input, p=p, dim=dim, eps=eps)

    @staticmethod
    @conservation_check_wrap
    def backward(ctx, R_out):
        return R_out, None, None, None


# ------------------------------------------------------------------------------- front-ends (same defaults)
def add2(input_a, input_b, inplace=False, epsilon=1e-8):
    return add2_tensors_fn.apply(input_a, input_b, inplace, epsilon)


def softmax(input, dim, dtype=None, temperature=1.0, inplace=False):
    return softmax_fn.apply(input, dim, dtype, temperature, inplace)


def linear_epsilon(input, weight, bias=None, epsilon=1e-6):
    return linear_epsilon_fn.apply(input, weight, bias, epsilon)


def matmul(input_a, input_b, inplace=False, epsilon=1e-8):
    return matmul_fn.apply(input_a, input_b, inplace, epsilon)


def rms_norm_identity(hidden_states, weight, variance_epsilon):
    return rms_norm_identity_fn.apply(hidden_states, weight, variance_epsilon)


def mul2(input_a, input_b, inplace=False):
    return mul2_fn.apply(input_a, input_b, inplace)


def mean(x, dim, keep_dim, epsilon=1e-6):
    return mean_fn.apply(x, dim, keep_dim, epsilon)


def layer_norm(hidden_states, weight, bias, variance_epsilon):
    return layer_norm_grad_fn.apply(hidden_states, weight, bias, variance_epsilon)


def normalize(input, p=2.0, dim=1, eps=1e-12, out=None):
    return normalize_identity_fn.apply(input, p, dim, eps)
