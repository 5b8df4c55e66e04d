"""Fused autograd Functions used by the patched forwards (lxt_amd.efficient.patches): each wraps one
forward kernel + the matching LRP-rule backward kernel of liblrp_hip.so.  These have no counterpart
file in the reference -- there the same arithmetic is spread over ATen ops; see the kernel
inventory in SURVEY.md 2.3 (K1-K8)."""
import torch
from torch.autograd import Function

from .. import ops


class RMSNormFn(Function):
    """K2: y = w' * x * rsqrt(mean(x^2)+eps) ; backward = identity rule (rsqrt detached):
    G_x = G_y * w' * rstd.   ref: lxt/efficient/patches.py:111-123, lxt/efficient/models/gemma3.py:11-12"""

    @staticmethod
    def forward(ctx, x, weight, eps, w_offset):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        y, rstd = ops.add_rmsnorm_fwd(x2, None, weight, eps, w_offset)
        ctx.save_for_backward(weight, rstd)
        ctx.w_offset = w_offset
        return y.view(shp)

    @staticmethod
    def backward(ctx, gy):
        weight, rstd = ctx.saved_tensors
        shp = gy.shape
        g2 = gy.reshape(-1, shp[-1]).contiguous()
        gx = torch.empty_like(g2)
        ops.rmsnorm_bwd_add2(None, g2, weight, rstd, None, None, gx, None, None, ctx.w_offset, 0.0, 0.0)
        return gx.view(shp), None, None, None


class LayerNormFn(Function):
    """K7: LayerNorm with std detached.  ref: lxt/efficient/patches.py:126-142"""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, _, rstd = ops.layernorm_fwd(x, weight, bias, eps)
        ctx.save_for_backward(weight, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        weight, rstd = ctx.saved_tensors
        return ops.layernorm_bwd(gy, None, weight, rstd, 0.0), None, None, None


class GatedActFn(Function):
    """K3: m = act(g) * u with the identity rule on act and the uniform rule on the product:
    G_g = 1/2 G_m u act(g)/(g+1e-10) ; G_u = 1/2 G_m act(g).   ref: lxt/efficient/patches.py:145-157"""

    @staticmethod
    def forward(ctx, g, u, act):
        shp = g.shape
        g2, u2 = g.reshape(-1, shp[-1]).contiguous(), u.reshape(-1, shp[-1]).contiguous()
        ctx.save_for_backward(g2, u2)
        ctx.act = act
        return ops.gated_act_fwd(g2, u2, None, act).view(shp)

    @staticmethod
    def backward(ctx, gm):
        g2, u2 = ctx.saved_tensors
        shp = gm.shape
        gm2 = gm.reshape(-1, shp[-1]).contiguous()
        Ag, Au = torch.empty_like(g2), torch.empty_like(u2)
        ops.gated_act_bwd(gm2, g2, u2, Ag, Au, 1e-10, 0.0, ctx.act)
        return Ag.view(shp), Au.view(shp), None


class FusedGatedMLPFn(Function):
    """The whole gated MLP of a decoder layer on the fused-epilogue GEMMs (bf16; the sequence lxt_amd.engine.LlamaLRP issues): gate/up GEMM over the
    INTERLEAVED fused weight with m = act(g) (*) u formed in its epilogue, down GEMM; backward: the down-projection dgrad with the gated rule
    (identity rule on act, uniform rule on the product: G_g = 1/2 G_m u act(g)/(g + 1e-10), G_u = 1/2 G_m act(g)) in its epilogue, then ONE gate/up
    dgrad over the fused weight.  ref: lxt/efficient/patches.py:145-157.  Wgu: ops.interleave_gate_up(gate.weight, up.weight); Wd: down.weight or a
    pitch-padded view of a copy (patches._fused_mlp_weights).  Long-K operands m [M, I] and Agu [M, 2 I] get a row pitch that is no multiple of
    4 KiB (engine.pitch_pad: +5 ... 14 % on those GEMMs)."""

    @staticmethod
    def forward(ctx, x, Wgu, Wd, act):
        from ..engine import pitch_pad
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if x2.stride(-1) != 1 or x2.stride(0) != x2.shape[1]:
            x2 = x2.contiguous()
        M, I = x2.shape[0], Wd.shape[1]
        es = x2.element_size()
        gu = torch.empty(M, 2 * I, device=x.device, dtype=x.dtype)
        m = torch.empty(M, I + pitch_pad(I, es), device=x.device, dtype=x.dtype)[:, :I]
        # M = B S rows: the backward's coefficients are stashed in gu's place (ops.gemm_gated_fwd_coef: no exp / rcp in the backward, g and u never
        # stored); otherwise the GEMM + element-wise pair on the stored gate/up output
        ctx.coef = ops.gated_coef_ok(M, I, x2.shape[1], x2.stride(0), Wgu.stride(0), Wd.shape[0], Wd.stride(0), act, x2.dtype)
        if ctx.coef:
            ops.gemm_gated_fwd_coef(x2, Wgu, gu, m, 1e-10, 0.0, act)
        else:
            ops.gemm_gated_fwd(x2, Wgu, gu, m, act)
        y = ops.linear_fwd(m, Wd, out=torch.empty(M, Wd.shape[0], device=x.device, dtype=x.dtype))
        ctx.save_for_backward(Wgu, Wd, gu)
        ctx.act = act
        return y.view(*shp[:-1], Wd.shape[0])

    @staticmethod
    def backward(ctx, gy):
        from ..engine import pitch_pad
        Wgu, Wd, gu = ctx.saved_tensors
        shp = gy.shape
        g2 = gy.reshape(-1, shp[-1])
        if g2.stride(-1) != 1 or g2.stride(0) != g2.shape[1]:
            g2 = g2.contiguous()
        M, I2 = gu.shape
        Agu = torch.empty(M, I2 + pitch_pad(I2, gu.element_size()), device=gu.device, dtype=gu.dtype)[:, :I2]
        if ctx.coef:
            ops.gemm_gated_bwd_coef(g2, Wd, gu, Agu)
        else:
            ops.gemm_gated_bwd(g2, Wd, gu, Agu, 1e-10, 0.0, ctx.act)
        gx = ops.linear_dgrad(Agu, Wgu, out=torch.empty(M, Wgu.shape[1], device=gu.device, dtype=gu.dtype))
        return gx.view(*shp[:-1], Wgu.shape[1]), None, None, None


class RopeFn(Function):
    """HF's apply_rotary_pos_emb on ONE tensor x [B, H, S, d] (a transposed view of the token-major projection output) as one launch of
    lrp_rope_fwd, its ordinary gradient as one launch of lrp_rope_bwd (lxt.efficient leaves RoPE un-patched: constant cos / sin, plain gradient;
    HF's eager form is ~8 element-wise ATen launches each way).  cos / sin: fp32 [B*S, d], one row per token (patches._rope_tables)."""

    @staticmethod
    def forward(ctx, x, cos, sin):
        B, H, S, d = x.shape
        xt = x.transpose(1, 2)
        if not xt.is_contiguous():
            xt = xt.contiguous()
        x2 = xt.view(B * S, H * d)
        out = torch.empty_like(x2)
        ops.rope_fwd(x2, out, cos, sin, B * S, H, d)
        ctx.save_for_backward(cos, sin)
        ctx.meta = (B, H, S, d)
        return out.view(B, S, H, d).transpose(1, 2)

    @staticmethod
    def backward(ctx, g):
        cos, sin = ctx.saved_tensors
        B, H, S, d = ctx.meta
        gt = g.transpose(1, 2)
        if not gt.is_contiguous():
            gt = gt.contiguous()
        g2 = gt.view(B * S, H * d)
        A = torch.empty_like(g2)
        ops.rope_bwd(g2, None, None, A, cos, sin, B * S, H, d, 0.0, 0.0)
        return A.view(B, S, H, d).transpose(1, 2), None, None


class LinearFn(Function):
    """K1 (efficient form, eps = 0): z = x W^T + b, backward G_x = G_z W -- both from the STORED weight [out, in] (ops.linear_fwd /
    ops.linear_dgrad pick the kernel by row count and dtype; bf16 never makes a W^T copy).  `weight_t` is accepted for callers of the
    round-2 signature and ignored."""

    @staticmethod
    def forward(ctx, x, weight, bias, weight_t=None):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        z = ops.linear_fwd(x2, weight, bias)
        ctx.save_for_backward(weight)
        return z.view(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gz):
        (w,) = ctx.saved_tensors
        shp = gz.shape
        g2 = gz.reshape(-1, shp[-1])
        if g2.stride(-1) != 1:
            g2 = g2.contiguous()
        gx = ops.linear_dgrad(g2, w)
        return gx.view(*shp[:-1], w.shape[1]), None, None, None


def _kernel_head_dim(d, dtype):
    """smallest head dim the attention kernels are instantiated for that holds d (SigLIP's 72, Phi's 80/96 ...)"""
    for cand in ((16, 32, 64, 128, 256) if dtype == torch.float32 else (32, 64, 128, 256)):
        if d <= cand:
            return cand
    raise NotImplementedError(f"attention head_dim {d} > 256 is not supported")


class AttentionFn(Function):
    """K4: flash attention forward + the AttnLRP backward (softmax Prop. 3.1, uniform rule on both
    matmuls == the reference's divide_gradient(q,4),(k,4),(v,2); ref: lxt/efficient/patches.py:193-203).
    q [B,S,Hq,d], k/v [B,S,Hkv,d] (token-major views), returns o [B,S,Hq,d]."""

    @staticmethod
    def forward(ctx, q, k, v, scale, causal, window, cp, row_iv=None):
        B, S, Hq, d0 = q.shape
        Hkv = k.shape[2]
        d = _kernel_head_dim(d0, q.dtype)
        if d != d0:     # zero-pad the head dim to the next size the kernels are built for: scores, softmax and the
            q, k, v = (torch.nn.functional.pad(t, (0, d - d0)) for t in (q, k, v))       # real columns are unchanged
        q2, k2, v2 = (t.reshape(B * S, -1).contiguous() for t in (q, k, v))
        v_t = ops.transpose_heads(v2, B, S, Hkv, d) if ops.attn_needs_transposed(q2, d) else None
        o = torch.empty_like(q2)
        lse = torch.empty(B, Hq, S, device=q.device, dtype=torch.float32)
        ops.attn_fwd(q2, k2, v2, v_t, o, lse, B, S, Hq, Hkv, d, scale, causal, window, row_iv=row_iv)
        ctx.save_for_backward(q2, k2, v2, o, lse)
        ctx.meta = (B, S, Hq, Hkv, d, d0, scale, causal, window, cp, row_iv)
        return o.view(B, S, Hq, d)[..., :d0]

    @staticmethod
    def backward(ctx, go):
        q2, k2, v2, o, lse = ctx.saved_tensors
        B, S, Hq, Hkv, d, d0, scale, causal, window, cp, row_iv = ctx.meta
        rep = Hq // Hkv
        if d != d0:
            go = torch.nn.functional.pad(go, (0, d - d0))
        go2 = go.reshape(B * S, Hq * d).contiguous()
        Gho = torch.empty_like(go2)
        D = torch.empty(B, Hq, S, device=go.device, dtype=torch.float32)
        # AttnLRP: uniform rule halves the relevance into P.V; CP-LRP (cp=True) keeps all of it on V
        ops.attn_bwd_prep(go2, o, Gho, D, B, S, Hq, d, 0.0, 1.0 if cp else 0.5)
        need_t = ops.attn_needs_transposed(q2, d)
        q_t, Gho_t = (ops.transpose_heads(q2, B, S, Hq, d), ops.transpose_heads(Gho, B, S, Hq, d)) if need_t else (None, None)
        dk_h, dv_h = torch.empty_like(q2), torch.empty_like(q2)
        ops.attn_bwd_dkv(q2, k2, v2, q_t, Gho, Gho_t, lse, D, dk_h, dv_h, B, S, Hq, Hkv, d, scale, 0.0, 0.0, causal, window,
                         row_iv=row_iv)
        dv = ops.gqa_reduce(dv_h, torch.empty_like(v2), B * S, Hkv, rep, d)
        if cp:      # CP-LRP: q and k are detached (ref: lxt/efficient/patches.py:245-255)
            return None, None, dv.view(B, S, Hkv, d)[..., :d0], None, None, None, None, None
        k_t = ops.transpose_heads(k2, B, S, Hkv, d) if need_t else None
        dq = torch.empty_like(q2)
        ops.attn_bwd_dq(q2, k2, v2, k_t, Gho, lse, D, dq, B, S, Hq, Hkv, d, scale, 0.0, 0.0, causal, window, row_iv=row_iv)
        dk = ops.gqa_reduce(dk_h, torch.empty_like(k2), B * S, Hkv, rep, d)
        return (dq.view(B, S, Hq, d)[..., :d0], dk.view(B, S, Hkv, d)[..., :d0], dv.view(B, S, Hkv, d)[..., :d0],
                None, None, None, None, None)
