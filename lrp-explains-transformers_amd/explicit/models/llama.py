"""Explicit AttnLRP for HuggingFace Llama on the HIP kernels -- the drop-in counterpart of the reference's `lxt.explicit.models.llama`.

ref: lxt/explicit/models/llama.py -- the reference ships a vendored copy of modeling_llama with the rule sites spliced in and two composites,
`attnlrp` and `cp_lrp` (:83-105).  That file cannot be imported under transformers 5.x (SURVEY.md finding 9), and a user of THIS framework
holds an ordinary `transformers.LlamaForCausalLM`; so the same rule placement is attached to an unmodified HF instance, in place and removably:

  :83-93    attnlrp = Composite({nn.SiLU: IdentityRule, ProjSiluMultiplication: UniformRule, nn.Softmax: SoftmaxDT,
                                 AttentionValueMatmul: UniformEpsilonRule, nn.Linear: EpsilonRule})
  :95-105   cp_lrp  = Composite({nn.SiLU: StopRelevanceRule, ProjSiluMultiplication: EpsilonRule, nn.Softmax: StopRelevanceRule,
                                 AttentionValueMatmul: EpsilonRule, nn.Linear: EpsilonRule})
  :226-260  RoPE: lf.add2(lf.mul2(q, cos), lf.mul2(rotate_half(q), sin)), rotate_half through lf.mul2(x2, -1)
  :273-281  MLP: down(ProjSiluMultiplication(act(gate(x)), up(x)))
  :379-391  attention: lf.mul2(lf.matmul(q, k^T), 1/sqrt(d)); lf.add2(., mask); SoftmaxDT (fp32); AttentionValueMatmul(p, v)
  :427      LlamaRMSNorm -> RMSNormIdentity (relevance passes through)
  :481-488  residuals: lf.add2(residual, hidden_states)

Usage (mirrors the reference's `attnlrp.register(model)`):
    from lxt_amd.explicit.models import llama
    model = LlamaForCausalLM.from_pretrained(...).cuda().eval()
    llama.attnlrp.register(model)                     # in place; llama.attnlrp.remove() restores the model
    e = model.get_input_embeddings()(ids).detach().requires_grad_()
    logits = model(inputs_embeds=e, use_cache=False).logits
    logits[0, -1, idx].backward(logits[0, -1, idx])   # explicit protocol (ref examples/paper/llama.py:45-46): relevance = e.grad
    R_token = e.grad.sum(-1)

Where the arithmetic runs.  Every nn.Linear goes through `rules.EpsilonRule` -> `lf.linear_epsilon` (MFMA GEMMs, eps-scale and the final (*) input on
the HIP kernels).  The attention block -- lf.matmul, the constant scale, the mask add2, SoftmaxDT and the uniform-eps P.V rule -- is ONE autograd
Function on the fused flash kernels: the S x S scores are never materialised (the reference's explicit path holds four [heads, S, S] tensors
per layer), relevance enters as R_o, becomes G_o' = 1/2 R_o / (o + 1e-6) (the uniform-eps rule's own first step), runs through the dQ and dK/dV
kernels with the stabilisers of lf.matmul (1e-8) and of the mask add2 (1e-8) folded into dS, and leaves as q (*) G_q, k (*) G_k, v (*) G_v
(repeat_kv is a plain expand in the reference: autograd sums the group, = lrp_gqa_reduce).  RoPE is one Function on lrp_rope_fwd / lrp_rope_bwd
with the add2 stabiliser.  Masks: whatever HF hands over is reduced to per-row key intervals (efficient/patches._mask_plan), so padded batches
work as in the efficient path."""
import types

import torch
import torch.nn as nn
from torch.autograd import Function

from ... import ops
from ...efficient.functions import _kernel_head_dim
from ...efficient.patches import _mask_plan
from .. import functional as lf
from .. import rules
from ..core import Composite

EPS_ADD, EPS_QK, EPS_MASK, EPS_PV, EPS_LIN = 1e-8, 1e-8, 1e-8, 1e-6, 1e-8       # ref :90,:258-259,:379,:384,:88 (rule defaults)


class _RopeFn(Function):
    """x [rows, H*d] token-major; cos / sin fp32 [rows, d] (HF's tables, one row per token).  ref :226-260: relevance form of
    add2(mul2(x, cos), mul2(rotate_half(x), sin)) with constant cos / sin: s = R / (x_rot + eps); R_x = x (*) rope^T(s)."""

    @staticmethod
    def forward(ctx, x, cos, sin, n_heads, d, eps):
        rows = x.shape[0]
        out = torch.empty_like(x)
        ops.rope_fwd(x, out, cos, sin, rows, n_heads, d)
        ctx.save_for_backward(x, out, cos, sin)
        ctx.meta = (n_heads, d, eps)
        return out

    @staticmethod
    @lf.conservation_check_wrap
    def backward(ctx, R):
        x, out, cos, sin = ctx.saved_tensors
        n_heads, d, eps = ctx.meta
        s = ops.eps_scale(R.contiguous(), out, 1.0, eps, relevance=True)
        A = torch.empty_like(x)
        ops.rope_bwd(s, None, None, A, cos, sin, x.shape[0], n_heads, d, 0.0, 0.0)
        return ops.mul(A, x), None, None, None, None, None


class _AttentionFn(Function):
    """q [B*S, Hq*d], k / v [B*S, Hkv*d] (post-RoPE, token-major) -> o [B*S, Hq*d].  cp = False: the attnlrp placement (ref :379-391);
    cp = True: cp_lrp (softmax: StopRelevanceRule -> nothing to q, k; P.V: EpsilonRule -> all of R through V, ref :95-105)."""

    @staticmethod
    def forward(ctx, q, k, v, B, S, Hq, Hkv, d0, scale, causal, window, row_iv, cp):
        d = _kernel_head_dim(d0, q.dtype)
        if d != d0:
            q, k, v = (torch.nn.functional.pad(t.view(B * S, -1, d0), (0, d - d0)).reshape(B * S, -1) for t in (q, k, v))
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        v_t = ops.transpose_heads(v, B, S, Hkv, d) if ops.attn_needs_transposed(q, d) else None
        o = torch.empty_like(q)
        lse = torch.empty(B, Hq, S, device=q.device, dtype=torch.float32)
        ops.attn_fwd(q, k, v, v_t, o, lse, B, S, Hq, Hkv, d, scale, causal, window, row_iv=row_iv)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.meta = (B, S, Hq, Hkv, d, d0, scale, causal, window, row_iv, cp)
        return o.view(B * S, Hq, d)[..., :d0].reshape(B * S, Hq * d0) if d != d0 else o

    @staticmethod
    @lf.conservation_check_wrap
    def backward(ctx, Ro):
        q, k, v, o, lse = ctx.saved_tensors
        B, S, Hq, Hkv, d, d0, scale, causal, window, row_iv, cp = ctx.meta
        rep = Hq // Hkv
        if d != d0:
            Ro = torch.nn.functional.pad(Ro.reshape(B * S, Hq, d0), (0, d - d0)).reshape(B * S, Hq * d)
        Ro = Ro.contiguous()
        # uniform-eps rule on P.V: G_o' = R_o / (o + eps) / 2  (cp_lrp: EpsilonRule -- no halving; the rule's eps default is 1e-8 there)
        Gho = ops.eps_scale(Ro, o, 1.0, EPS_LIN, relevance=True) if cp else ops.eps_scale(Ro, o, 2.0, 2.0 * EPS_PV, relevance=True)
        D = (Gho.view(B, S, Hq, d).float() * o.view(B, S, Hq, d).float()).sum(-1).permute(0, 2, 1).contiguous()
        need_t = ops.attn_needs_transposed(q, d)
        q_t, Gho_t = (ops.transpose_heads(q, B, S, Hq, d), ops.transpose_heads(Gho, B, S, Hq, d)) if need_t else (None, None)
        dk_h, dv_h = torch.empty_like(q), torch.empty_like(q)
        em, eq = (0.0, 0.0) if cp else (EPS_MASK, EPS_QK)
        ops.attn_bwd_dkv(q, k, v, q_t, Gho, Gho_t, lse, D, dk_h, dv_h, B, S, Hq, Hkv, d, scale, em, eq, causal, window, row_iv=row_iv)
        Rv = ops.mul(ops.gqa_reduce(dv_h, torch.empty_like(v), B * S, Hkv, rep, d), v)
        Rq = Rk = None
        if not cp:
            k_t = ops.transpose_heads(k, B, S, Hkv, d) if need_t else None
            dq = torch.empty_like(q)
            ops.attn_bwd_dq(q, k, v, k_t, Gho, lse, D, dq, B, S, Hq, Hkv, d, scale, em, eq, causal, window, row_iv=row_iv)
            Rq = ops.mul(dq, q)
            Rk = ops.mul(ops.gqa_reduce(dk_h, torch.empty_like(k), B * S, Hkv, rep, d), k)
        if d != d0:
            cut = lambda t, H: t.view(B * S, H, d)[..., :d0].reshape(B * S, H * d0) if t is not None else None    # noqa: E731
            Rq, Rk, Rv = cut(Rq, Hq), cut(Rk, Hkv), cut(Rv, Hkv)
        return (Rq, Rk, Rv) + (None,) * 10


def _mul(a, b):
    return a * b


def _attention_forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, **kwargs):
    """LlamaAttention.forward with the explicit rule sites (ref :340-400); self._lxt_cp selects cp_lrp"""
    if past_key_values is not None:
        raise NotImplementedError("lxt_amd explicit Llama: kv cache is outside the explained path (call the model with use_cache=False)")
    B, S, _ = hidden_states.shape
    d = self.head_dim
    Hq, Hkv = self.config.num_attention_heads, self.config.num_key_value_heads
    q = self.q_proj(hidden_states).reshape(B * S, Hq * d)
    k = self.k_proj(hidden_states).reshape(B * S, Hkv * d)
    v = self.v_proj(hidden_states).reshape(B * S, Hkv * d)
    cos, sin = position_embeddings
    cos, sin = (t.expand(B, S, d).reshape(B * S, d).float().contiguous() for t in (cos, sin))
    cp = bool(getattr(self, "_lxt_cp", False))
    if cp:      # no relevance reaches q / k (softmax: StopRelevanceRule): plain rotation
        with torch.no_grad():
            q, k = _RopeFn.apply(q.detach(), cos, sin, Hq, d, 0.0), _RopeFn.apply(k.detach(), cos, sin, Hkv, d, 0.0)
    else:
        q, k = _RopeFn.apply(q, cos, sin, Hq, d, EPS_ADD), _RopeFn.apply(k, cos, sin, Hkv, d, EPS_ADD)
    window = int(kwargs.get("sliding_window") or 0)
    causal, window, row_iv = _mask_plan(attention_mask, S, self, 0 if window >= S else window)
    o = _AttentionFn.apply(q, k, v, B, S, Hq, Hkv, d, float(self.scaling), causal, window, row_iv, cp)
    return self.o_proj(o.view(B, S, Hq * d)), None


def _decoder_layer_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                           position_embeddings=None, **kwargs):
    """ref :470-490: the two residual sums are lf.add2"""
    residual = hidden_states
    hidden_states = self.input_layernorm(hidden_states)
    hidden_states, _ = self.self_attn(hidden_states=hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                                      past_key_values=past_key_values, use_cache=use_cache, position_embeddings=position_embeddings, **kwargs)
    hidden_states = lf.add2(residual, hidden_states, epsilon=EPS_ADD)
    residual = hidden_states
    hidden_states = self.post_attention_layernorm(hidden_states)
    hidden_states = self.mlp(hidden_states)
    return lf.add2(residual, hidden_states, epsilon=EPS_ADD)


def _mlp_forward(self, x):
    """ref :273-281: ProjSiluMultiplication -> UniformRule (attnlrp) / EpsilonRule (cp_lrp); the activation module itself was swapped by the
    Composite (IdentityRule / StopRelevanceRule)"""
    g, u = self.act_fn(self.gate_proj(x)), self.up_proj(x)
    m = rules.epsilon_lrp(_mul, EPS_LIN, g, u) if getattr(self, "_lxt_cp", False) else rules.uniform_rule_fn.apply(_mul, g, u)
    return self.down_proj(m)


def _rmsnorm_forward(self, hidden_states):
    """LlamaRMSNorm -> RMSNormIdentity (ref :427, lxt/explicit/modules.py:35-45)"""
    return lf.rms_norm_identity(hidden_states, self.weight, self.variance_epsilon)


class LlamaComposite:
    """`attnlrp` / `cp_lrp` of the reference: a Composite for the module-level rules plus the function-level rule sites of the vendored model
    file, attached to the instances of ONE model (other Llama instances in the process are untouched)."""

    def __init__(self, cp=False):
        self.cp = cp
        self._composite = None
        self._patched = []

    def layer_map(self):
        from transformers.activations import SiLUActivation
        act_rule = rules.StopRelevanceRule if self.cp else rules.IdentityRule
        return {nn.SiLU: act_rule, SiLUActivation: act_rule, nn.Linear: rules.EpsilonRule}

    def register(self, model, dummy_inputs=None, tracer=None, verbose=False, no_grad=True):
        from transformers.models.llama import modeling_llama as ml
        if self._composite is not None:
            raise RuntimeError("this composite is already registered on a model: call remove() first")
        for m in model.modules():
            fwd = None
            if isinstance(m, ml.LlamaAttention):
                fwd = _attention_forward
            elif isinstance(m, ml.LlamaDecoderLayer):
                fwd = _decoder_layer_forward
            elif isinstance(m, ml.LlamaMLP):
                fwd = _mlp_forward
            elif isinstance(m, ml.LlamaRMSNorm):
                fwd = _rmsnorm_forward
            if fwd is not None:
                m.forward = types.MethodType(fwd, m)
                m._lxt_cp = self.cp
                self._patched.append(m)
        self._composite = Composite(self.layer_map())
        self._composite.register(model, verbose=verbose, no_grad=no_grad)
        return model

    def remove(self):
        if self._composite is not None:
            self._composite.remove()
        for m in self._patched:
            m.__dict__.pop("forward", None)
            m.__dict__.pop("_lxt_cp", None)
        self._composite, self._patched = None, []

    def context(self, model, **kwargs):
        from contextlib import contextmanager

        @contextmanager
        def cm():
            try:
                yield self.register(model, **kwargs)
            finally:
                self.remove()
        return cm()


attnlrp = LlamaComposite(cp=False)
cp_lrp = LlamaComposite(cp=True)
