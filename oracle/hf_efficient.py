"""Oracle for the DROP-IN path: a CPU restatement of lxt.efficient's gradient modifiers applied to
a HuggingFace model at INSTANCE level (no class patching, so it can run in the same process as the
HIP-backed lxt_amd.efficient.monkey_patch without interfering).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
  ref: lxt/efficient/rules.py:88-127      identity_rule_implicit / divide_gradient
  ref: lxt/efficient/patches.py:111-220   rms_norm / layer_norm / gated_mlp / mlp / attention / dropout
  ref: lxt/efficient/models/gemma3.py:11-12, models/gpt2.py:11-15, models/bert.py:581,790,806
with plain torch autograd Functions (runs in fp32 or fp64 on the host).  Pinned by
tests/golden/make_golden.py: same per-token relevance as the real lxt.efficient on Llama (monkey_patch)
and as the real reference primitives on HF BERT / Gemma3-text (custom patch maps), fixtures
tests/golden/{llama_*,bert_base,gemma3_tiny}.npz.
"""
import types

import torch
import torch.nn as nn
from torch.autograd import Function


class _IdentityRule(Function):
    @staticmethod
    def forward(ctx, fn, x, eps=1e-10):
        y = fn(x)
        ctx.save_for_backward(y / (x + eps))
        return y

    @staticmethod
    def backward(ctx, g):
        return None, ctx.saved_tensors[0] * g, None


class _DivideGradient(Function):
    @staticmethod
    def forward(ctx, x, factor):
        ctx.factor = factor
        return x

    @staticmethod
    def backward(ctx, g):
        return g / ctx.factor, None


def identity_rule_implicit(fn, x):
    return _IdentityRule.apply(fn, x)


def divide_gradient(x, factor=2):
    return _DivideGradient.apply(x, factor)


def _rms_forward(self, hidden_states):
    dt = hidden_states.dtype
    h = hidden_states.to(torch.float32) if dt != torch.float64 else hidden_states
    var = h.pow(2).mean(-1, keepdim=True)
    eps = getattr(self, "variance_epsilon", None)
    eps = self.eps if eps is None else eps
    h = h * torch.rsqrt(var + eps).detach()
    return self.weight * h.to(dt)


def _gemma3_rms_forward(self, x):
    xf = x.float() if x.dtype != torch.float64 else x
    out = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps).detach()
    out = out * (1.0 + (self.weight.float() if x.dtype != torch.float64 else self.weight))
    return out.type_as(x)


def _layer_norm_forward(self, x):
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    std = (var + self.eps).sqrt()
    y = (x - mean) / std.detach()
    if self.weight is not None:
        y = y * self.weight
    if self.bias is not None:
        y = y + self.bias
    return y


def _gated_mlp_forward(self, x):
    gate = identity_rule_implicit(self.act_fn, self.gate_proj(x))
    return self.down_proj(divide_gradient(gate * self.up_proj(x), 2))


def _bert_intermediate_forward(self, hidden_states):
    return identity_rule_implicit(self.intermediate_act_fn, self.dense(hidden_states))


def _bert_pooler_forward(self, hidden_states):
    return identity_rule_implicit(self.activation, self.dense(hidden_states[:, 0]))


def _gpt2_mlp_forward(self, hidden_states):
    return self.c_proj(identity_rule_implicit(self.act, self.c_fc(hidden_states)))


def _dropout_forward(self, x):
    return x


def _linear_forward(self, x):
    # nn.Linear is untouched by the reference (plain F.linear); pinned per instance so the oracle model stays
    # on the host even after lxt_amd's class-level Linear patch (HIP GEMM) is active in the same process
    return torch.nn.functional.linear(x, self.weight, self.bias)


def lrp_eager_attention(module, query, key, value, attention_mask=None, scaling=None, dropout=0.0, **kwargs):
    """HF eager attention with the AttnLRP factors: grad(q)/4, grad(k)/4, grad(v)/2."""
    query, key, value = divide_gradient(query, 4), divide_gradient(key, 4), divide_gradient(value, 2)
    rep = query.shape[1] // key.shape[1]
    if rep > 1:
        key = key.repeat_interleave(rep, dim=1)
        value = value.repeat_interleave(rep, dim=1)
    if scaling is None:
        scaling = query.shape[-1] ** -0.5
    w = torch.matmul(query, key.transpose(2, 3)) * scaling
    if attention_mask is not None:
        w = w + attention_mask[:, :, :, : key.shape[-2]]
    elif getattr(module, "is_causal", False) and query.shape[2] > 1:
        S = query.shape[2]
        w = w.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=w.device).tril(), float("-inf"))
    sm_dtype = torch.float64 if w.dtype == torch.float64 else torch.float32
    w = torch.softmax(w, dim=-1, dtype=sm_dtype).to(query.dtype)
    out = torch.matmul(w, value)
    return out.transpose(1, 2).contiguous(), w


def cp_eager_attention(module, query, key, value, attention_mask=None, scaling=None, dropout=0.0, **kwargs):
    """CP-LRP (ref: lxt/efficient/patches.py:245-255): q and k detached, no uniform-rule factors"""
    class _Plain:   # reuse the arithmetic of lrp_eager_attention without its divide_gradient calls
        pass
    query, key = query.detach(), key.detach()
    rep = query.shape[1] // key.shape[1]
    if rep > 1:
        key = key.repeat_interleave(rep, dim=1)
        value = value.repeat_interleave(rep, dim=1)
    if scaling is None:
        scaling = query.shape[-1] ** -0.5
    w = torch.matmul(query, key.transpose(2, 3)) * scaling
    if attention_mask is not None:
        w = w + attention_mask[:, :, :, : key.shape[-2]]
    elif getattr(module, "is_causal", False) and query.shape[2] > 1:
        S = query.shape[2]
        w = w.masked_fill(~torch.ones(S, S, dtype=torch.bool, device=w.device).tril(), float("-inf"))
    sm_dtype = torch.float64 if w.dtype == torch.float64 else torch.float32
    w = torch.softmax(w, dim=-1, dtype=sm_dtype).to(query.dtype)
    return torch.matmul(w, value).transpose(1, 2).contiguous(), w


def _cp_gated_mlp_forward(self, x):
    """ref: lxt/efficient/patches.py:269-280"""
    gate = self.act_fn(self.gate_proj(x).detach())
    return self.down_proj(gate * self.up_proj(x))


def patch_instance(model, variant="attnlrp", skip=(), attn_configs=("text_config", "vision_config")):
    """Apply the efficient AttnLRP (variant="attnlrp") or CP-LRP (variant="cp") rule placement to ONE
    model instance (instance-level forwards + a private attention interface).  Returns the model.
    skip: module-name prefixes left untouched (except nn.Linear, pinned to the host everywhere) -- the reference's
    gemma3 map patches nothing inside the SigLIP tower (ref: lxt/efficient/models/gemma3.py:14-19);
    attn_configs: which sub-configs get the LRP attention (the reference reaches SigLIP's attention only through the
    process-wide sdpa registry entry, not with eager: ref lxt/efficient/patches.py:171-190)."""
    from transformers import AttentionInterface, AttentionMaskInterface
    from transformers.masking_utils import eager_mask
    cp = variant == "cp"
    name_attn = "lrp_oracle_cp" if cp else "lrp_oracle"
    AttentionInterface.register(name_attn, cp_eager_attention if cp else lrp_eager_attention)
    AttentionMaskInterface.register(name_attn, eager_mask)      # HF builds the per-layer (causal / sliding) additive masks
    subs = {k: getattr(model.config, k) for k in ("text_config", "vision_config") if hasattr(model.config, k)}
    keep = {k: c._attn_implementation for k, c in subs.items() if k not in attn_configs}
    model.config._attn_implementation = name_attn            # (HF propagates the top-level setting to the sub-configs)
    for k, c in subs.items():
        c._attn_implementation = keep.get(k, name_attn)
    for mname, m in model.named_modules():
        name = type(m).__name__
        if isinstance(m, nn.Linear):
            m.forward = types.MethodType(_linear_forward, m)
        elif any(mname.startswith(pre) for pre in skip):
            continue
        elif isinstance(m, nn.Dropout):
            m.forward = types.MethodType(_dropout_forward, m)
        elif isinstance(m, nn.LayerNorm):
            m.forward = types.MethodType(_layer_norm_forward, m)
        elif name == "Gemma3RMSNorm":
            m.forward = types.MethodType(_gemma3_rms_forward, m)
        elif name.endswith("RMSNorm"):
            m.forward = types.MethodType(_rms_forward, m)
        elif name.endswith("MLP") and hasattr(m, "gate_proj") and hasattr(m, "act_fn"):
            m.forward = types.MethodType(_cp_gated_mlp_forward if cp else _gated_mlp_forward, m)
        elif name == "GPT2MLP":
            m.forward = types.MethodType(_gpt2_mlp_forward, m)
        elif name == "BertIntermediate":
            m.forward = types.MethodType(_bert_intermediate_forward, m)
        elif name == "BertPooler":
            m.forward = types.MethodType(_bert_pooler_forward, m)
    for p in model.parameters():
        p.requires_grad_(False)
    return model


def explain_causal_lm(model, ids, target=None):
    """user protocol of docs/source/quickstart.rst:120-141 on a patched causal LM (batch 1)"""
    e = model.get_input_embeddings()(ids[None]).detach().requires_grad_()
    logits = model(inputs_embeds=e, use_cache=False).logits
    last = logits[0, -1]
    idx = int(last.argmax()) if target is None else int(target)
    last[idx].backward()
    R = (e * e.grad)[0]
    return dict(idx=idx, logit=float(last[idx]), R_tok=R.sum(-1).detach(), R_emb=R.detach(), logits_last=last.detach())


def explain_classifier(model, ids, target=None):
    """BERT-style sequence classifier (docs/source/quickstart.rst:199-211): explain the max logit"""
    e = model.get_input_embeddings()(ids[None]).detach().requires_grad_()
    logits = model(inputs_embeds=e).logits[0]
    idx = int(logits.argmax()) if target is None else int(target)
    logits[idx].backward()
    R = (e * e.grad)[0]
    return dict(idx=idx, logit=float(logits[idx]), R_tok=R.sum(-1).detach(), R_emb=R.detach(), logits=logits.detach())
