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


class DecoderLayerFn(Function):
    """ONE Llama-type decoder layer (pre-norm attention block + pre-norm gated MLP, both with residuals) as the launch sequence of
    lxt_amd.engine.LlamaLRP's dense layer -- for an adopted bf16 HF model at M = B S rows: the two RMSNorms and both residual sums inside the GEMM
    epilogues (K1n: norm weights folded into the fused [q;k;v] and gate/up weights by patches._fused_layer_weights), one fused QKV GEMM, the
    gated rule as a coefficient stash, flash attention; backward: 7 GEMM launches, the dQ kernel forms D and applies RoPE's backward in its store,
    dK's group sum carries RoPE's backward too -- no attn_bwd_prep / rope_bwd / norm passes.  Rules as lxt.efficient places them (ref
    lxt/efficient/patches.py:111-123 rms_norm_forward, :145-157 gated_mlp_forward, :193-203 wrap_attention_forward; HF modeling_llama's
    LlamaDecoderLayer.forward for the wiring).  h [B, S, H]; cos / sin fp32 [S, d] (one table for every prompt: plain causal, un-padded batches);
    rstd1 fp32 [B S]: 1 / rms of h's rows (from the previous layer's epilogue, or computed by the caller).  Returns (h_out, rstd of h_out's rows)."""

    @staticmethod
    def forward(ctx, h, rstd1, Wqkv, Wo, Wgu, Wd, cos, sin, meta):
        from ..engine import pitch_pad
        nq, nk, d, eps, act, scale = meta
        B, S, H = h.shape
        M, I = B * S, Wd.shape[1]
        nqk, nqkv = (nq + nk) * d, (nq + 2 * nk) * d
        dev, dt, es = h.device, h.dtype, h.element_size()
        new = lambda r, c, pad=0: torch.empty(r, c + pad, device=dev, dtype=dt)[:, :c]          # noqa: E731
        h2 = h.reshape(M, H)
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        qkv = new(M, nqkv)
        if ops.gemm_nt_rs_rope_ok(h2, Wqkv, qkv, S, nqk, d):
            qkr = ops.gemm_nt_rs_rope(h2, Wqkv, rstd1, cos, sin, qkv, S, nqk, d)[:, :nqk]      # RoPE in the GEMM's epilogue
        else:
            qkr = ops.rope_fwd(ops.gemm_nt_rs(h2, Wqkv, rstd1, qkv), new(M, nqk), cos, sin, S, nq + nk, d)
        q, k, v = qkr[:, : nq * d], qkr[:, nq * d:], qkv[:, nqk:]
        o, lse = new(M, nq * d), torch.empty(B, nq, S, device=dev, dtype=torch.float32)
        ops.attn_fwd(q, k, v, None, o, lse, B, S, nq, nk, d, scale, True, 0)
        ssq = torch.empty(H // 64, M, device=dev, dtype=torch.float32)
        h1 = ops.gemm_res_ssq(o, Wo, h2, new(M, H), ssq)
        rstd2 = ops.rms_rstd(ssq, M, H, eps, torch.empty(M, device=dev, dtype=torch.float32))
        coef, m = ops.gemm_gated_fwd_coef(h1, Wgu, new(M, 2 * I), new(M, I, pitch_pad(I, es)), 1e-10, 0.0, act, rs=rstd2)
        out = ops.gemm_res_ssq(m, Wd, h1, new(M, H), ssq)
        rstd_out = ops.rms_rstd(ssq, M, H, eps, torch.empty(M, device=dev, dtype=torch.float32))
        ctx.save_for_backward(rstd1, rstd2, qkv, qkr, o, lse, coef, Wqkv, Wo, Wgu, Wd, cos, sin)
        ctx.meta = (B, S, H, I, nq, nk, d, scale)
        ctx.mark_non_differentiable(rstd_out)
        return out.view(B, S, H), rstd_out

    @staticmethod
    def backward(ctx, gy, _g_rstd):
        from ..engine import pitch_pad
        rstd1, rstd2, qkv, qkr, o, lse, coef, Wqkv, Wo, Wgu, Wd, cos, sin = ctx.saved_tensors
        B, S, H, I, nq, nk, d, scale = ctx.meta
        M, rep = B * S, nq // nk
        nqk, nqkv = (nq + nk) * d, (nq + 2 * nk) * d
        dev, dt, es = gy.device, gy.dtype, gy.element_size()
        new = lambda r, c, pad=0: torch.empty(r, c + pad, device=dev, dtype=dt)[:, :c]          # noqa: E731
        G = gy.reshape(M, H)
        if not G.is_contiguous():
            G = G.contiguous()
        Agu = ops.gemm_gated_bwd_coef(G, Wd, coef, new(M, 2 * I, pitch_pad(2 * I, es)))
        Gs1 = ops.gemm_nn_rs_res(Agu, Wgu, rstd2, G, new(M, H))
        half = ops.const_rows(M, 0.5, dev)
        Gho = ops.gemm_nn_rs(Gs1, Wo, half, new(M, nq * d))                                       # 1/2 (uniform rule on P.V): exact row scale
        q, k, v = qkr[:, : nq * d], qkr[:, nq * d:], qkv[:, nqk:]
        D = torch.empty(B, nq, S, device=dev, dtype=torch.float32)
        Aqkv = new(M, nqkv, 64 if (nqkv * es) % 4096 == 0 else 0)
        ops.attn_bwd_dq_d(q, k, v, Gho, o, lse, D, Aqkv[:, : nq * d], B, S, nq, nk, d, scale, rope=(cos, sin))
        dk_h, dv_h = new(M, nq * d), new(M, nq * d)
        ops.attn_bwd_dkv(q, k, v, None, Gho, None, lse, D, dk_h, dv_h, B, S, nq, nk, d, scale, 0.0, 0.0)
        ops.gqa_reduce_rope(dk_h, Aqkv[:, nq * d: nqk], M, S, nk, rep, d, cos, sin)
        ops.gqa_reduce(dv_h, Aqkv[:, nqk:], M, nk, rep, d)
        Gh = ops.gemm_nn_rs_res(Aqkv, Wqkv, rstd1, Gs1, new(M, H))
        return Gh.view(B, S, H), None, None, None, None, None, None, None, None


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
