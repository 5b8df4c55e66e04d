"""CP-LRP attention for torch.nn.MultiheadAttention (vision transformers): relevance flows through the value path only,
the softmax output is a constant (ref: lxt/explicit/special.py:33-140).  Projections and both attention contractions run
on liblrp_hip.so's batched GEMM; the softmax is the lrp_softmax_fwd kernel."""
import math

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import ops


@torch.no_grad()
def _prepare_key_padding_mask(mask, attn_mask, query):
    """ref: special.py:8-19"""
    assert mask.ndim > 1
    if mask.ndim == 2:
        b, k_len = mask.shape
        mask = mask.view(b, 1, 1, k_len)
    return F._canonical_mask(mask, "key_padding_mask", F._none_or_dtype(attn_mask), "attn_mask", query.dtype)


@torch.no_grad()
def _prepare_attn_mask(mask, query):
    """ref: special.py:21-31"""
    assert mask.ndim >= 2
    if mask.ndim == 3:
        mask = mask.view(query.shape)
    return F._canonical_mask(mask, "attn_mask", None, "", query.dtype, False)


class pv_epsilon_fn(Function):
    """y = P V with P constant; epsilon rule on V: R_v = V (*) (P^T (R / (y + eps)))   -- what
    rules.epsilon_lrp(torch.matmul, 1e-6, attention.detach(), v) computes (ref: special.py:121, rules.py:170-222)"""

    @staticmethod
    def forward(ctx, p, v, epsilon):
        p, v = p.contiguous(), v.contiguous()
        y = ops.gemm_nt(p, ops.transpose(v))                      # [.., S, S] x [.., d, S]^T -> [.., S, d]
        ctx.save_for_backward(p, v, y)
        ctx.epsilon = epsilon
        return y

    @staticmethod
    def backward(ctx, R_out):
        p, v, y = ctx.saved_tensors
        s = ops.eps_scale(R_out.contiguous(), y, 1.0, ctx.epsilon, relevance=True)
        g = ops.gemm_nt(ops.transpose(p), ops.transpose(s))       # P^T s : [.., S, S] x [.., d, S]^T -> [.., S, d]
        return None, ops.mul(g, v), None


def multi_head_attention_cp(query, key, value, batch_first, num_heads, head_dim, q_proj_weight, bias_q, k_proj_weight, bias_k,
                            v_proj, out_proj, key_padding_mask=None, need_weights=True, attn_mask=None, average_attn_weights=True):
    """same signature and return convention as the reference (special.py:33-140)"""
    if batch_first is False:
        query, key, value = query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1)
    B, Sq, E = query.shape
    Sk = value.shape[1]
    with torch.no_grad():
        q = ops.gemm_nt(query.detach().reshape(B * Sq, E).contiguous(), q_proj_weight.detach(), bias_q)
        k = ops.gemm_nt(key.detach().reshape(B * Sk, E).contiguous(), k_proj_weight.detach(), bias_k)
    v = v_proj(value)
    q = q.view(B, Sq, num_heads, head_dim).permute(0, 2, 1, 3).contiguous()
    k = k.view(B, Sk, num_heads, head_dim).permute(0, 2, 1, 3).contiguous()
    v = v.view(B, Sk, num_heads, head_dim).permute(0, 2, 1, 3)
    with torch.no_grad():
        logits = ops.gemm_nt(q, k)                                 # [B, H, Sq, Sk]
        if key_padding_mask is not None or attn_mask is not None:
            mask = torch.zeros_like(logits)
            if key_padding_mask is not None:
                mask += _prepare_key_padding_mask(key_padding_mask, attn_mask, q)
            if attn_mask is not None:
                mask += _prepare_attn_mask(attn_mask, q)
            # softmax_fwd scales its input: (s + m*sqrt(d)) / sqrt(d) = s/sqrt(d) + m
            logits = logits + mask * math.sqrt(head_dim)
        attention = ops.softmax_fwd(logits.contiguous(), 1.0 / math.sqrt(head_dim))
    y = pv_epsilon_fn.apply(attention, v, 1e-6)
    y = y.permute(0, 2, 1, 3).reshape(B, Sq, E)
    out = out_proj(y)
    if batch_first is False:
        out = out.transpose(0, 1)
    if need_weights and average_attn_weights:
        return out, attention.mean(dim=1)
    if need_weights:
        return out, attention
    return out, None
