"""Drop-in rule modules (ref: lxt/explicit/modules.py:13-54) and the module-swap initialisers used by
Composite (ref: modules.py:127-214)."""
import torch
import torch.nn as nn

from . import functional as lf
from . import special as ls


class SoftmaxDT(nn.Softmax):
    def __init__(self, dim, dtype=None, temperature=1.0, inplace=False, **kwargs):
        super().__init__(dim)
        self.inplace, self.dtype, self.temperature = inplace, dtype, temperature

    def forward(self, inputs):
        return lf.softmax(inputs, self.dim, self.dtype, self.temperature, self.inplace)


class LinearEpsilon(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None, epsilon=1e-6, **kwargs):
        super().__init__(in_features, out_features, bias, device, dtype)
        self.epsilon = epsilon

    def forward(self, inputs):
        return lf.linear_epsilon(inputs, self.weight, self.bias, self.epsilon)


class RMSNormIdentity(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        return lf.rms_norm_identity(hidden_states, self.weight, self.variance_epsilon)


class LayerNormEpsilon(nn.LayerNorm):
    def forward(self, x):
        return lf.layer_norm(x, self.weight, self.bias, self.eps)


class _LinearProjection(nn.Module):
    """plain Linear holding (possibly sliced) weight / bias tensors so that a rule can be attached to it
    (ref: modules.py:60-85); forward and input-gradient on the HIP GEMM"""

    def __init__(self, weight, bias):
        super().__init__()
        self.weight = weight
        self.bias = bias

    def forward(self, x):
        from ..efficient.functions import LinearFn
        w = self.weight if self.weight.is_contiguous() else self.weight.contiguous()
        return LinearFn.apply(x, w.detach(), self.bias.detach() if self.bias is not None else None)


class LinearInProjection(_LinearProjection):
    pass


class LinearOutProjection(_LinearProjection):
    pass


class MultiheadAttention_CP(nn.Module):
    """CP-LRP replacement of torch.nn.MultiheadAttention (ref: modules.py:87-123): attach rules to .v_proj / .out_proj"""

    def __init__(self):
        super().__init__()
        self.q_proj_weight = None
        self.k_proj_weight = None
        self.v_proj = LinearInProjection(None, None)
        self.out_proj = LinearOutProjection(None, None)
        self.embed_dim = self.num_heads = self.head_dim = self.batch_first = None
        self.bias_q = self.bias_k = None

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None, average_attn_weights=True,
                is_causal=False):
        assert is_causal is False        # not supported by the reference either
        return ls.multi_head_attention_cp(query, key, value, self.batch_first, self.num_heads, self.head_dim, self.q_proj_weight,
                                          self.bias_q, self.k_proj_weight, self.bias_k, self.v_proj, self.out_proj,
                                          key_padding_mask, need_weights, attn_mask, average_attn_weights)


def initialize_MHA(original, replacement):
    """ref: modules.py:173-206"""
    new = replacement()
    E = original.embed_dim
    if not original._qkv_same_embed_dim:
        new.q_proj_weight, new.k_proj_weight, new.v_proj.weight = original.q_proj_weight, original.k_proj_weight, original.v_proj_weight
    else:
        w = original.in_proj_weight
        new.q_proj_weight, new.k_proj_weight, new.v_proj.weight = w[:E], w[E: 2 * E], w[2 * E: 3 * E]
    if original.in_proj_bias is not None:
        b = original.in_proj_bias
        new.bias_q, new.bias_k, new.v_proj.bias = b[:E], b[E: 2 * E], b[2 * E: 3 * E]
    if original.bias_k is not None:
        raise NotImplementedError("add_bias_kv=True is not supported yet.")
    new.out_proj.weight, new.out_proj.bias = original.out_proj.weight, original.out_proj.bias
    new.embed_dim, new.num_heads, new.head_dim, new.batch_first = original.embed_dim, original.num_heads, original.head_dim, original.batch_first
    return new


def _share_params(new, old, names):
    for n in names:
        p = getattr(old, n, None)
        if p is not None:
            setattr(new, n, p)
    return new


def initialize_linear_epsilon(original, rule_cls):
    new = rule_cls(original.in_features, original.out_features, original.bias is not None,
                   device=original.weight.device, dtype=original.weight.dtype)
    return _share_params(new, original, ("weight", "bias"))


def initialize_rms_norm_identity(original, rule_cls):
    eps = getattr(original, "variance_epsilon", getattr(original, "eps", 1e-6))
    new = rule_cls(original.weight.shape[0], eps)
    return _share_params(new, original, ("weight",))


def initialize_layer_norm_epsilon(original, rule_cls):
    new = rule_cls(original.normalized_shape, original.eps, original.elementwise_affine, original.bias is not None,
                   device=original.weight.device if original.weight is not None else None)
    return _share_params(new, original, ("weight", "bias"))


def initialize_softmax_dt(original, rule_cls):
    return rule_cls(original.dim)


INIT_MODULE_MAPPING = {
    LinearEpsilon: initialize_linear_epsilon,
    RMSNormIdentity: initialize_rms_norm_identity,
    LayerNormEpsilon: initialize_layer_norm_epsilon,
    SoftmaxDT: initialize_softmax_dt,
    MultiheadAttention_CP: initialize_MHA,
}
