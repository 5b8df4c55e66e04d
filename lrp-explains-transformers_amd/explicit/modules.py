"""Drop-in rule modules (ref: lxt/explicit/modules.py:13-54) and the module-swap initialisers used by
Composite (ref: modules.py:127-214)."""
import torch
import torch.nn as nn

from . import functional as lf


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
}
