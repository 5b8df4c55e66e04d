"""Gamma rule for nn.Linear and patch-embedding nn.Conv2d in the gradient x input framework (vision transformers:
ref docs/source/quickstart.rst:376-384 pairs the vit_torch map with zennit's `Gamma` through `monkey_patch_zennit`,
lxt/efficient/zennit_patches.py:32-62).

PARITY UNPINNED: zennit is a third-party dependency of the reference (un-pinned in setup.py, not vendored, absent from
this image), so the rule below restates zennit's published generalised Gamma (rules.py, 0.5.x) and is checked against
oracle/rules.py:gamma_linear_gxi only -- there is no reference fixture for it.  All eight contractions (four modified
forward passes, four input-gradients) run on liblrp_hip.so's GEMM; the masks / stabilisers are element-wise glue.

Usage mirrors a zennit composite:   comp = GammaComposite([(nn.Conv2d, 0.25), (nn.Linear, 0.05)]);  comp.register(model)
... forward / backward / (x * x.grad) ...;  comp.remove()"""
import types

import torch
from torch import nn
from torch.autograd import Function

from .. import ops


def _stab(x, eps):
    return x + ((x == 0).to(x.dtype) + x.sign()) * eps


class GammaLinearFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gamma, eps, cache):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        z = ops.linear_fwd(x2, weight, bias)            # the same dispatch as the un-ruled LinearFn: identical forward values
        ctx.save_for_backward(x2, z)
        ctx.meta = (weight, bias, gamma, eps, cache, shp)
        return z.view(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, G):
        x, z = ctx.saved_tensors
        weight, bias, gamma, eps, cache, shp = ctx.meta
        key = (weight.data_ptr(), weight._version, float(gamma))
        if cache.get("key") != key:          # modified weights and their transposes, once per (weight, gamma)
            Wp = (weight + gamma * weight.clamp(min=0)).contiguous()
            Wm = (weight + gamma * weight.clamp(max=0)).contiguous()
            cache.update(key=key, Wp=Wp, Wm=Wm, WpT=ops.transpose(Wp), WmT=ops.transpose(Wm))
        Wp, Wm, WpT, WmT = cache["Wp"], cache["Wm"], cache["WpT"], cache["WmT"]
        bp = bm = None
        if bias is not None:
            bp, bm = bias + gamma * bias.clamp(min=0), bias + gamma * bias.clamp(max=0)
        xp, xm = x.clamp(min=0), x.clamp(max=0)
        f32 = torch.float32                    # the ratios R / z+- are formed in fp32 (fp32 GEMM outputs) whatever the storage dtype
        zpos = ops.gemm_nt(xp, Wp, bp, out_dtype=f32) + ops.gemm_nt(xm, Wm, out_dtype=f32)
        zneg = ops.gemm_nt(xp, Wm, bm, out_dtype=f32) + ops.gemm_nt(xm, Wp, out_dtype=f32)
        zf = z.float()
        R = G.reshape(z.shape).float() * zf
        sp = torch.where(zf > 0, R / _stab(zpos, eps), torch.zeros_like(R)).to(x.dtype).contiguous()
        sn = torch.where(zf < 0, R / _stab(zneg, eps), torch.zeros_like(R)).to(x.dtype).contiguous()
        R_in = xp * (ops.gemm_nt(sp, WpT) + ops.gemm_nt(sn, WmT)) + xm * (ops.gemm_nt(sp, WmT) + ops.gemm_nt(sn, WpT))
        return (R_in / _stab(x, 1e-10)).view(shp), None, None, None, None, None


def _linear_forward(self, x):
    b = self.bias.detach() if self.bias is not None else None
    return GammaLinearFn.apply(x, self.weight.detach(), b, self._lrp_gamma, self._lrp_gamma_eps, self._lrp_gamma_cache)


def _conv_forward(self, x):
    kh, kw = self.kernel_size
    if tuple(self.stride) != (kh, kw) or self.groups != 1 or tuple(self.dilation) != (1, 1) or self.padding not in ((0, 0), "valid", 0):
        raise NotImplementedError("lxt_amd Gamma rule: only patch-embedding convolutions (stride == kernel, no padding) are supported")
    B, C, Hh, Ww = x.shape
    gh, gw = Hh // kh, Ww // kw
    patches = x[:, :, : gh * kh, : gw * kw].reshape(B, C, gh, kh, gw, kw).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, C * kh * kw)
    w2 = self.weight.detach().reshape(self.out_channels, -1)
    b = self.bias.detach() if self.bias is not None else None
    y = GammaLinearFn.apply(patches, w2, b, self._lrp_gamma, self._lrp_gamma_eps, self._lrp_gamma_cache)
    return y.view(B, gh, gw, self.out_channels).permute(0, 3, 1, 2)


class GammaComposite:
    """instance-level Gamma rule registration, zennit-composite style: layer_map = [(module type, gamma), ...]"""

    def __init__(self, layer_map, epsilon=1e-6):
        self.layer_map, self.epsilon, self._patched = list(layer_map), epsilon, []

    def register(self, model):
        for m in model.modules():
            for typ, gamma in self.layer_map:
                if isinstance(m, typ) and isinstance(m, (nn.Linear, nn.Conv2d)):
                    m._lrp_gamma, m._lrp_gamma_eps = float(gamma), float(self.epsilon)
                    if not hasattr(m, "_lrp_gamma_cache"):
                        m._lrp_gamma_cache = {}
                    m.forward = types.MethodType(_linear_forward if isinstance(m, nn.Linear) else _conv_forward, m)
                    self._patched.append(m)
                    break
        return model

    def remove(self):
        for m in self._patched:
            if "forward" in m.__dict__:
                del m.__dict__["forward"]
        self._patched = []
