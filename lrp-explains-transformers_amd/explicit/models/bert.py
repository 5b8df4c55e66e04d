"""Explicit AttnLRP for HuggingFace BERT (BASELINE config 2 in lxt.explicit semantics), on the HIP kernels.

ref: lxt/explicit/models/bert.py -- the rule placement restated on an unmodified HF model instance:
  :60-65    attnlrp = Composite({nn.ReLU / nn.Tanh / GELUActivation: IdentityRule, nn.Linear: EpsilonRule})
  :204,:396 every LayerNorm is lm.LayerNormEpsilon (lf.layer_norm, eps 1e-6 on y)
  :249-253  embeddings: lf.add2(inputs_embeds, token_type.detach()), lf.add2(., position), LayerNorm
  :338-373  attention: lf.matmul(q, k^T), lf.mul2(., 1/sqrt(d)), lf.add2(., mask), lf.softmax, lf.matmul(p, v)
            -- BOTH contractions use lf.matmul (R / (2 O + 1e-8)); the explicit Llama composite uses the uniform-eps rule for P.V
  :396 ff.  residuals: LayerNorm(lf.add2(dense(h), input))

Usage (mirrors the reference's `attnlrp.register(model)`):
    from lxt_amd.explicit.models import bert
    model = BertForSequenceClassification(...).cuda().eval()
    bert.attnlrp.register(model)                     # in place; bert.attnlrp.remove() restores the model
    e = model.get_input_embeddings()(ids).detach().requires_grad_()
    logits = model(inputs_embeds=e).logits
    logits[0, idx].backward(logits[0, idx])          # explicit protocol: seed with the logit, relevance = e.grad
Scores are materialised ([B, heads, S, S] through lf.matmul / lf.softmax on the HIP GEMM and row kernels): the explicit rule
API is defined on materialised operands; at BERT's S = 128 that is 0.8 MB per layer.
"""
import math
import types

import torch
import torch.nn as nn

from .. import functional as lf
from .. import rules
from ..core import Composite

ATTN_NAME = "lxt_amd_explicit"


def explicit_attention_forward(module, query, key, value, attention_mask=None, scaling=None, dropout=0.0, **kwargs):
    """HF attention-interface function: q [B,H,S,d], k/v [B,H,S,d] -> (out [B,S,H,d], None); ref :338-373"""
    d = query.shape[-1]
    scale = float(scaling) if scaling is not None else 1.0 / math.sqrt(d)
    s = lf.matmul(query, key.transpose(-1, -2))
    s = lf.mul2(s, scale)
    if attention_mask is None:       # no padding: the reference's BertModel builds the all-zero extended mask and add2's it
        attention_mask = torch.zeros(1, dtype=s.dtype, device=s.device)
    elif attention_mask.dtype == torch.bool:   # a boolean "may attend" mask -> the additive form the reference adds (ref :346)
        attention_mask = torch.zeros_like(attention_mask, dtype=s.dtype).masked_fill(~attention_mask, torch.finfo(s.dtype).min)
    s = lf.add2(s, attention_mask.to(s.dtype).expand_as(s))
    p = lf.softmax(s, dim=-1)
    c = lf.matmul(p, value)
    return c.transpose(1, 2).contiguous(), None


def _embeddings_forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None, past_key_values_length=0):
    """ref :226-254"""
    if inputs_embeds is None:
        inputs_embeds = self.word_embeddings(input_ids)
    B, S = inputs_embeds.shape[:2]
    if position_ids is None:
        position_ids = self.position_ids[:, past_key_values_length: S + past_key_values_length]
    if token_type_ids is None:
        token_type_ids = torch.zeros(B, S, dtype=torch.long, device=inputs_embeds.device)
    tt = self.token_type_embeddings(token_type_ids)
    emb = lf.add2(inputs_embeds, tt.detach())
    pos = self.position_embeddings(position_ids).expand(B, S, -1)
    emb = lf.add2(emb, pos)
    return self.dropout(self.LayerNorm(emb))


def _residual_forward(self, hidden_states, input_tensor):
    """BertSelfOutput / BertOutput; ref :396-400, :449-453"""
    hidden_states = self.dropout(self.dense(hidden_states))
    return self.LayerNorm(lf.add2(hidden_states, input_tensor))


def _layer_norm_forward(self, x):
    """lm.LayerNormEpsilon.forward on an nn.LayerNorm instance; ref lxt/explicit/modules.py:48-54"""
    return lf.layer_norm(x, self.weight, self.bias, self.eps)


def register_interfaces(attention_fn=None):
    """Register the explicit attention function AND a mask builder under ATTN_NAME.  transformers >= 4.53 builds the attention
    mask per `_attn_implementation` through AttentionMaskInterface; a name without a mask function gets attention_mask=None, i.e.
    padded batches would silently attend to pad tokens.  The eager builder yields the additive [B,1,S,S] mask that the reference
    feeds to lf.add2 (ref: lxt/explicit/models/bert.py:346)."""
    from transformers import AttentionInterface
    AttentionInterface.register(ATTN_NAME, attention_fn or explicit_attention_forward)
    try:
        from transformers.masking_utils import AttentionMaskInterface, eager_mask
    except ImportError:          # older transformers: the model builds the extended additive mask itself
        return
    AttentionMaskInterface.register(ATTN_NAME, eager_mask)


class BertAttnLRP:
    """`attnlrp` of the reference (a Composite) plus the function-level rules its vendored model file carries."""

    def __init__(self):
        self._composite = None
        self._patched = []
        self._config = None

    def register(self, model, verbose=False, no_grad=True):
        from transformers.activations import GELUActivation
        from transformers.models.bert import modeling_bert as mb
        register_interfaces()
        self._config = (model.config, model.config._attn_implementation)
        model.config._attn_implementation = ATTN_NAME
        for m in model.modules():
            fwd = None
            if isinstance(m, mb.BertEmbeddings):
                fwd = _embeddings_forward
            elif isinstance(m, (mb.BertSelfOutput, mb.BertOutput)):
                fwd = _residual_forward
            elif isinstance(m, nn.LayerNorm):
                fwd = _layer_norm_forward
            if fwd is not None:
                m.forward = types.MethodType(fwd, m)
                self._patched.append(m)
        self._composite = Composite({nn.ReLU: rules.IdentityRule, nn.Tanh: rules.IdentityRule, nn.Linear: rules.EpsilonRule,
                                     GELUActivation: rules.IdentityRule, nn.GELU: rules.IdentityRule})
        self._composite.register(model, verbose=verbose, no_grad=no_grad)
        return model

    def remove(self):
        if self._composite is not None:
            self._composite.remove()
        for m in self._patched:
            m.__dict__.pop("forward", None)
        if self._config is not None:
            self._config[0]._attn_implementation = self._config[1]
        self._composite, self._patched, self._config = None, [], None


attnlrp = BertAttnLRP()
