"""Fused AttnLRP driver for Gemma-3 text decoders (BASELINE config 4): forward + relevance backward of `lxt.efficient` on
Gemma3ForCausalLM as straight sequences of liblrp_hip launches -- no autograd, no module hooks.

What differs from the Llama driver (engine.py), with the reference lines that fix the behaviour:
  * (1 + w) RMSNorm with the rstd detached (ref lxt/efficient/models/gemma3.py:11-12 `gemma3_norm`, HF Gemma3RMSNorm.forward):
    the norm kernels' `w_offset = 1`; the backward is the row-constant scale G (1 + w) rstd;
  * FOUR norms per layer (HF Gemma3DecoderLayer.forward): input -> attention -> post-attention norm -> residual add;
    pre-feed-forward norm -> gated MLP -> post-feed-forward norm -> residual add;
  * per-head q / k RMSNorm (head_dim-wide rows) in front of RoPE (HF Gemma3Attention.forward);
  * gelu-tanh gated MLP under the same identity + uniform rules (ref gemma3.py:15 -> patches.py:145-157 `gated_mlp_forward`);
  * sliding-window layers (5 local : 1 global in the released checkpoints) with their own rotary base, attention scale
    query_pre_attn_scalar ** -0.5, embeddings scaled by sqrt(hidden);
  * attention factors as everywhere in lxt.efficient (ref patches.py:193-203: q, k / 4, v / 2) -- inside the attention kernels.
The reference defines Gemma-3 for the efficient mode only (there is no lxt.explicit Gemma-3), and so does this driver.
Weights are held once (forward layout; dgrads run the NN GEMM on the stored weight), gate/up interleaved for the fused GEMM epilogues,
activations live in the same tagged arena as the Llama driver's.
"""
import torch

from . import ops
from .engine import EFFICIENT, LlamaLRP, pitch_pad, weight_pitch_pad

_STATIC_ROPE = ("default", "linear")


def config_from_hf(hf_cfg):
    """HF Gemma3TextConfig (or the text_config of a Gemma3Config) -> engine cfg; unsupported features are refused loudly"""
    hf_cfg = getattr(hf_cfg, "text_config", hf_cfg)
    mt = getattr(hf_cfg, "model_type", "")
    if mt not in ("gemma3_text", "gemma3"):
        raise NotImplementedError(f"Gemma3LRP drives Gemma-3 text decoders only (model_type={mt!r})")
    if getattr(hf_cfg, "attention_bias", False):
        raise NotImplementedError("Gemma3LRP: attention_bias = True is not supported by the fused driver (use monkey_patch)")
    if getattr(hf_cfg, "attn_logit_softcapping", None) or getattr(hf_cfg, "final_logit_softcapping", None):
        raise NotImplementedError("Gemma3LRP: logit soft-capping is not supported by the fused driver (use monkey_patch)")
    if not getattr(hf_cfg, "use_bidirectional_attention", False) is False:
        raise NotImplementedError("Gemma3LRP: bidirectional attention is not supported")
    act = getattr(hf_cfg, "hidden_activation", getattr(hf_cfg, "hidden_act", "gelu_pytorch_tanh"))
    if act not in ("gelu_pytorch_tanh", "gelu_tanh", "silu"):
        raise NotImplementedError(f"Gemma3LRP: activation {act!r}")
    rope = {}
    rp_all = getattr(hf_cfg, "rope_parameters", None)
    for lt in sorted(set(hf_cfg.layer_types)):
        if isinstance(rp_all, dict) and isinstance(rp_all.get(lt), dict):
            rp = rp_all[lt]                                   # transformers 5: one parameter set per layer type
        elif lt == "sliding_attention":                       # transformers 4.x schema: rope_local_base_freq for the local layers,
            rp = dict(rope_type="default", rope_theta=getattr(hf_cfg, "rope_local_base_freq", 10000.0))
        else:                                                 # rope_theta (+ rope_scaling, key 'rope_type' or legacy 'type') for the global ones
            rs = dict(getattr(hf_cfg, "rope_scaling", None) or {})
            rp = dict(rs, rope_type=rs.get("rope_type", rs.get("type", "default")), rope_theta=getattr(hf_cfg, "rope_theta", 1e6))
        kind = rp.get("rope_type", "default")
        if kind not in _STATIC_ROPE:
            raise NotImplementedError(f"Gemma3LRP: rope_type {kind!r} (layer type {lt!r}) is not supported by the fused driver")
        d = hf_cfg.head_dim
        inv = 1.0 / (float(rp["rope_theta"]) ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
        if kind == "linear":                                  # HF `_compute_linear_scaling_rope_parameters`: inv_freq / factor, no post-scale
            inv = inv / float(rp["factor"])
        rope[lt] = (inv.float().cpu(), 1.0)
    return dict(hidden=hf_cfg.hidden_size, inter=hf_cfg.intermediate_size, n_layers=hf_cfg.num_hidden_layers,
                n_heads=hf_cfg.num_attention_heads, n_kv=hf_cfg.num_key_value_heads, head_dim=hf_cfg.head_dim,
                vocab=hf_cfg.vocab_size, rms_eps=float(hf_cfg.rms_norm_eps), act="silu" if act == "silu" else "gelu_tanh",
                scale=float(hf_cfg.query_pre_attn_scalar) ** -0.5, layer_types=list(hf_cfg.layer_types),
                window=int(hf_cfg.sliding_window), rope=rope, embed_scale=float(hf_cfg.hidden_size) ** 0.5)


def weights_from_hf(model):
    """plain (cfg, W) view of a HF Gemma3ForCausalLM / the language model of a Gemma3ForConditionalGeneration"""
    lm = model.model if hasattr(model, "lm_head") else model
    m = getattr(lm, "language_model", lm)
    m = getattr(m, "model", m) if not hasattr(m, "layers") else m
    W = dict(embed=m.embed_tokens.weight.detach(), norm=m.norm.weight.detach(), lm_head=model.lm_head.weight.detach(), layers=[])
    for L in m.layers:
        a, mlp = L.self_attn, L.mlp
        W["layers"].append(dict(ln_in=L.input_layernorm.weight.detach(), ln_pa=L.post_attention_layernorm.weight.detach(),
                                ln_pf=L.pre_feedforward_layernorm.weight.detach(), ln_pff=L.post_feedforward_layernorm.weight.detach(),
                                qn=a.q_norm.weight.detach(), kn=a.k_norm.weight.detach(),
                                wq=a.q_proj.weight.detach(), wk=a.k_proj.weight.detach(), wv=a.v_proj.weight.detach(),
                                wo=a.o_proj.weight.detach(), wg=mlp.gate_proj.weight.detach(), wu=mlp.up_proj.weight.detach(),
                                wd=mlp.down_proj.weight.detach()))
    cfg = config_from_hf(model.config)
    return cfg, W


class Gemma3LRP:
    """explain(input_ids) -> dict(idx, logit, R_tok, logits): AttnLRP (lxt.efficient) token relevances of a Gemma-3 text decoder"""

    def __init__(self, cfg, W, dtype=torch.bfloat16, device="cuda", max_seq=4096, sparse_top=True):
        if not torch.cuda.is_available():
            raise RuntimeError("Gemma3LRP needs a HIP device: the LRP kernels have no CPU fallback")
        self.cfg, self.dtype, self.device = dict(cfg), dtype, torch.device(device)
        # top-layer sparsity (as LlamaLRP): only the explained position's logit is used, so above the last layer's attention every row but one
        # per prompt is dead in the forward and carries zero gradient in the backward -- its o-projection, norms and MLP run on B rows instead
        # of B S, its attention on one query row per prompt (result-preserving: test_gemma3_engine_top_layer_sparsity_equals_dense)
        self.sparse_top = bool(sparse_top)
        self.eps, self.act = dict(EFFICIENT), cfg["act"]
        self.eps_g = self.eps["act"]
        dev = self.device
        H, I, nq, nk, d, V = cfg["hidden"], cfg["inter"], cfg["n_heads"], cfg["n_kv"], cfg["head_dim"], cfg["vocab"]
        nqkv = (nq + 2 * nk) * d
        es = torch.empty(0, dtype=dtype).element_size()
        up = lambda n: (n + 63) // 64 * 64                                   # noqa: E731
        tied = W["lm_head"].data_ptr() == W["embed"].data_ptr()
        wpad = weight_pitch_pad(H, es, 2 * I)              # gate/up weight: stored row pitch off the 1-KiB grid (its dgrad strides over the rows)
        per_layer = 4 * up(H) + 2 * up(d) + up(nqkv * H) + up(H * nq * d) + up(2 * I * (H + wpad)) + up(H * (I + pitch_pad(I, es)))
        total = (1 if tied else 2) * up(V * H) + up(H) + len(W["layers"]) * per_layer
        self.flat = torch.empty(total, device=dev, dtype=dtype)          # ONE buffer: a multi-GPU start-up is a single broadcast
        cursor = [0]

        def take(*shape):
            n = 1
            for s_ in shape:
                n *= s_
            v = self.flat[cursor[0]: cursor[0] + n].view(*shape)
            cursor[0] += up(n)
            return v

        def put(dst, *srcs):
            o = 0
            for t in srcs:
                dst[o: o + t.shape[0]].copy_(t.to(device=dev, dtype=dtype, non_blocking=True))
                o += t.shape[0]
            return dst

        self.embed = put(take(V, H), W["embed"])
        self.lm_head = self.embed if tied else put(take(V, H), W["lm_head"])
        self.norm = put(take(H), W["norm"])
        self.layers = []
        for L in W["layers"]:
            pad = pitch_pad(I, es)
            self.layers.append(dict(
                ln_in=put(take(H), L["ln_in"]), ln_pa=put(take(H), L["ln_pa"]), ln_pf=put(take(H), L["ln_pf"]), ln_pff=put(take(H), L["ln_pff"]),
                qn=put(take(d), L["qn"]), kn=put(take(d), L["kn"]),
                wqkv=put(take(nqkv, H), L["wq"], L["wk"], L["wv"]), wo=put(take(H, nq * d), L["wo"]),
                wgu=ops.interleave_gate_up(L["wg"].to(device=dev, dtype=dtype), L["wu"].to(device=dev, dtype=dtype), out=take(2 * I, H + wpad)[:, :H]),
                wd=put(take(H, I + pad)[:, :I], L["wd"])))
        self.attn_t = ops.attn_needs_transposed(self.embed, d)
        # rotary tables per layer type (HF hands cos / sin to the layers in the model dtype: keep that rounding, store fp32)
        self.rope = {}
        for lt, (inv, att) in cfg["rope"].items():
            fr = torch.arange(max_seq, dtype=torch.float32)[:, None] * inv[None, :]
            e = torch.cat((fr, fr), dim=-1)
            self.rope[lt] = ((e.cos() * att).to(dtype).to(torch.float32).to(dev).contiguous(),
                             (e.sin() * att).to(dtype).to(torch.float32).to(dev).contiguous())
        self.embed_scale = torch.tensor(cfg["embed_scale"], dtype=dtype)   # HF: the scale itself is rounded to the model dtype
        self.max_seq = max_seq
        self._arena = None
        torch.cuda.synchronize(dev)

    @classmethod
    def from_hf(cls, model, **kw):
        cfg, W = weights_from_hf(model)
        kw.setdefault("dtype", next(model.parameters()).dtype)
        return cls(cfg, W, **kw)

    def release(self):
        self._arena = None

    def _window(self, li):
        return self.cfg["window"] if self.cfg["layer_types"][li] == "sliding_attention" else 0

    def _attn_mask(self, li, row_iv):
        """-> (causal, window, row intervals) of layer li.  row_iv is None / a (lo, hi) pair (left-padded text batches: causal + window stay
        structural) or a dict {"global": (lo, hi), "local": (lo, hi)} of COMPLETE per-row key intervals (image + text prompts: tokens of one
        image attend to each other in both directions, HF `create_masks_for_vision_model`; the intervals then carry causality and the
        sliding window themselves, see engine_gemma3_mm.mm_row_intervals)"""
        if isinstance(row_iv, dict):
            return False, 0, row_iv["local" if self.cfg["layer_types"][li] == "sliding_attention" else "global"]
        return True, self._window(li), row_iv

    # ---------------------------------------------------------------------------------------------
    def forward(self, emb, B, S, row_iv=None):
        c = self.cfg
        H, I, d, nq, nk = c["hidden"], c["inter"], c["head_dim"], c["n_heads"], c["n_kv"]
        M, dt, dev = B * S, self.dtype, self.device
        nqd, nkd, nqkv = nq * d, nk * d, (nq + 2 * nk) * d
        if self._arena is None:
            self._arena = LlamaLRP._Arena(dev)
        ar = self._arena
        new = lambda tag, *s: ar.get(tag, s, dt)  # noqa: E731
        wide = lambda tag, r, c_: ar.get(tag, (r, c_), dt, pad=pitch_pad(c_, emb.element_size()))  # noqa: E731
        f32 = lambda tag, *s: ar.get(tag, s, torch.float32)  # noqa: E731
        eps = c["rms_eps"]
        stash = []
        h_prev, branch, nxt = emb, None, None
        site = ops.sandwich_norm_ok(emb)      # Gemma-3's norms / q-k norm / RoPE fused per site (ops.SITE_FUSION; results bit-identical)
        for li, Lw in enumerate(self.layers):
            st = {}
            cos, sin = self.rope[c["layer_types"][li]]
            # input norm (fused with the previous layer's residual add: h = h1_prev + post_ff_norm(dn_prev); with the site kernels the previous
            # layer's tail has produced h, x and rstd1 already)
            if nxt is not None:
                h, x, st["rstd1"] = nxt
            else:
                x, st["rstd1"] = new("x", M, H), f32(("rstd1", li), M)
                if branch is None:
                    h = h_prev
                    ops.add_rmsnorm_fwd(h_prev, None, Lw["ln_in"], eps, 1.0, y=x, rstd=st["rstd1"])
                else:
                    h = new(("h", li & 1), M, H)
                    ops.add_rmsnorm_fwd(h_prev, branch, Lw["ln_in"], eps, 1.0, hsum_out=h, y=x, rstd=st["rstd1"])
            qkv = ops.linear_fwd(x, Lw["wqkv"], out=new(("qkv", li), M, nqkv))
            # per-head q / k norm straight out of the fused projection output (strided rows), then RoPE with this layer type's table
            st["rstd_q"], st["rstd_k"] = f32(("rstd_q", li), M * nq), f32(("rstd_k", li), M * nk)
            if site:      # one pass: q / k norm + RoPE (bit-identical to the four launches below)
                qr, kr = ops.qk_norm_rope_fwd(qkv, Lw["qn"], Lw["kn"], new(("qr", li), M, nqd), new(("kr", li), M, nkd), st["rstd_q"], st["rstd_k"],
                                              cos, sin, S, nq, nk, d, eps, 1.0)
            else:
                qn, kn = new("qn", M, nqd), new("kn", M, nkd)
                ops.head_rmsnorm_fwd(qkv[:, :nqd], Lw["qn"], qn, st["rstd_q"], nq, d, eps, 1.0)
                ops.head_rmsnorm_fwd(qkv[:, nqd: nqd + nkd], Lw["kn"], kn, st["rstd_k"], nk, d, eps, 1.0)
                qr = ops.rope_fwd(qn, new(("qr", li), M, nqd), cos, sin, S, nq, d)
                kr = ops.rope_fwd(kn, new(("kr", li), M, nkd), cos, sin, S, nk, d)
            v = qkv[:, nqd + nkd:]
            v_t = ops.transpose_heads(v, B, S, nk, d) if self.attn_t else None
            o, lse = new(("o", li), M, nqd), f32(("lse", li), B, nq, S)
            causal, win, iv = self._attn_mask(li, row_iv)
            if self.sparse_top and li == len(self.layers) - 1:
                # ---- the top layer above its attention: one row per prompt
                ops.attn_fwd(qr, kr, v, v_t, o, lse, B, S, nq, nk, d, c["scale"], causal, win, q_begin=S - 1, row_iv=iv)
                last = torch.arange(B, device=dev) * S + (S - 1)
                o_l, h_l = o.index_select(0, last), h.index_select(0, last)
                a_l = ops.linear_fwd(o_l, Lw["wo"], out=new("a_l", B, H))
                pa_l, rstd_pa_l = ops.add_rmsnorm_fwd(a_l, None, Lw["ln_pa"], eps, 1.0)
                h1_l = new("h1_l", B, H)
                x2_l, rstd2_l = ops.add_rmsnorm_fwd(h_l, pa_l, Lw["ln_pf"], eps, 1.0, hsum_out=h1_l)
                gu_l, m_l = ops.gemm_gated_fwd(x2_l, Lw["wgu"], new("gu_l", B, 2 * I), new("m_l", B, I), self.act)
                dn_l = ops.linear_fwd(m_l, Lw["wd"], out=new("dn_l", B, H))
                pff_l, rstd_pff_l = ops.add_rmsnorm_fwd(dn_l, None, Lw["ln_pff"], eps, 1.0)
                st.update(top=True, qkv=qkv, qr=qr, kr=kr, lse=lse, o_l=o_l, gu_l=gu_l, rstd_pa_l=rstd_pa_l, rstd2_l=rstd2_l, rstd_pff_l=rstd_pff_l)
                stash.append(st)
                h_prev, branch = h1_l, pff_l
                break
            ops.attn_fwd(qr, kr, v, v_t, o, lse, B, S, nq, nk, d, c["scale"], causal, win, row_iv=iv)
            a = ops.linear_fwd(o, Lw["wo"], out=new("a", M, H))
            # post-attention norm, residual add, pre-feed-forward norm
            st["rstd_pa"] = f32(("rstd_pa", li), M)
            h1, x2, st["rstd2"] = new(("h1", li & 1), M, H), new("x2", M, H), f32(("rstd2", li), M)
            if site:      # one pass (the normed branch never goes to memory)
                ops.sandwich_norm_fwd(a, h, Lw["ln_pa"], Lw["ln_pf"], eps, 1.0, h1, x2, st["rstd_pa"], st["rstd2"])
            else:
                pa = new("pa", M, H)
                ops.add_rmsnorm_fwd(a, None, Lw["ln_pa"], eps, 1.0, y=pa, rstd=st["rstd_pa"])
                ops.add_rmsnorm_fwd(h, pa, Lw["ln_pf"], eps, 1.0, hsum_out=h1, y=x2, rstd=st["rstd2"])
            coef = ops.gated_coef_ok(M, I, H, H, Lw["wgu"].stride(0), H, Lw["wd"].stride(0), self.act, dt)
            if coef:      # gated rules inside the two GEMMs: the backward's coefficients are stashed in gu's place (ops.gemm_gated_fwd_coef)
                gu, m = ops.gemm_gated_fwd_coef(x2, Lw["wgu"], new(("gu", li), M, 2 * I), wide("m", M, I), self.eps_g, self.eps["lin"], self.act)
            else:
                gu, m = ops.gemm_gated_fwd(x2, Lw["wgu"], new(("gu", li), M, 2 * I), wide("m", M, I), self.act)
            dn = ops.linear_fwd(m, Lw["wd"], out=new("dn", M, H))
            st["rstd_pff"] = f32(("rstd_pff", li), M)
            st.update(qkv=qkv, qr=qr, kr=kr, o=o, lse=lse, gu=gu, coef=coef)
            stash.append(st)
            if site and li + 1 < len(self.layers):
                # post-feed-forward norm, residual add and the NEXT layer's input norm in one pass
                nxt = (new(("h", (li + 1) & 1), M, H), new("x", M, H), f32(("rstd1", li + 1), M))
                ops.sandwich_norm_fwd(dn, h1, Lw["ln_pff"], self.layers[li + 1]["ln_in"], eps, 1.0, nxt[0], nxt[1], st["rstd_pff"], nxt[2])
                continue
            nxt = None
            pff = new("pff", M, H)
            ops.add_rmsnorm_fwd(dn, None, Lw["ln_pff"], eps, 1.0, y=pff, rstd=st["rstd_pff"])
            h_prev, branch = h1, pff
        last = torch.arange(B, device=dev) * S + (S - 1)
        if h_prev.shape[0] == B and stash and stash[-1].get("top", False):
            h_last, b_last = h_prev, branch                        # (the sparse top layer left one row per prompt already)
        else:
            h_last = h_prev.index_select(0, last)
            b_last = branch.index_select(0, last) if branch is not None else None
        hL_last = new("hL_last", B, H)
        xn, rstd_f = ops.add_rmsnorm_fwd(h_last, b_last, self.norm, eps, 1.0, hsum_out=hL_last)
        logits = ops.linear_fwd(xn, self.lm_head, out=f32("logits", B, c["vocab"]))
        return dict(stash=stash, last=last, rstd_f=rstd_f, logits=logits, row_iv=row_iv, site=site)

    # ---------------------------------------------------------------------------------------------
    def backward(self, fw, idx, B, S):
        c, E = self.cfg, self.eps
        H, I, d, nq, nk = c["hidden"], c["inter"], c["head_dim"], c["n_heads"], c["n_kv"]
        M, rep, dt, dev = B * S, nq // nk, self.dtype, self.device
        nqd, nkd, nqkv = nq * d, nk * d, (nq + 2 * nk) * d
        ar = self._arena
        new = lambda tag, *s: ar.get(tag, s, dt)  # noqa: E731
        wide = lambda tag, r, c_: ar.get(tag, (r, c_), dt, pad=pitch_pad(c_, torch.empty(0, dtype=dt).element_size()))  # noqa: E731
        f32 = lambda tag, *s: ar.get(tag, s, torch.float32)  # noqa: E731
        zeros = lambda tag, *s: ar.get(tag, s, dt, zero=True)  # noqa: E731
        # LM head (gradient of the explained logit) + final (1 + w) norm on the one explained row of each prompt, scattered into [M, H]
        Gh_last = ops.head_seed(self.lm_head, fw["logits"], idx, self.norm, fw["rstd_f"], new("Gh_last", B, H), 1.0, E["lin"])
        top = bool(fw["stash"]) and fw["stash"][-1].get("top", False)
        Gs = None if top else zeros(("Gs", len(self.layers) & 1), M, H).index_copy_(0, fw["last"], Gh_last)       # gradient w.r.t. h_L = h1 + pff
        site, Gdn = fw["site"], None
        for li in range(len(self.layers) - 1, -1, -1):
            Lw, st = self.layers[li], fw["stash"][li]
            cos, sin = self.rope[c["layer_types"][li]]
            causal, win, iv = self._attn_mask(li, fw["row_iv"])
            q_begin = 0
            if st.get("top", False):
                # ---- one row per prompt through the post-feed-forward norm, the MLP, the two norms around the residual add and the o-projection; then
                # scatter into the dense operands of the attention backward (zero fill + row scatter; D's live column S - 1 is a strided scatter)
                Gdn_l, Gs1_l, Ga_l = new("Gdn_l", B, H), new("Gs1_l", B, H), new("Ga_l", B, H)
                ops.rmsnorm_bwd_add2(None, Gh_last, Lw["ln_pff"], st["rstd_pff_l"], None, None, Gdn_l, None, None, 1.0, 0.0, 0.0)
                Agu_l = ops.gemm_gated_bwd(Gdn_l, Lw["wd"], st["gu_l"], new("Agu_l", B, 2 * I), self.eps_g, E["lin"], self.act)
                Gx2_l = ops.linear_dgrad(Agu_l, Lw["wgu"], out=new("Gx2_l", B, H))
                ops.rmsnorm_bwd_add2(Gh_last, Gx2_l, Lw["ln_pf"], st["rstd2_l"], None, None, Gs1_l, None, None, 1.0, 0.0, 0.0)
                ops.rmsnorm_bwd_add2(None, Gs1_l, Lw["ln_pa"], st["rstd_pa_l"], None, None, Ga_l, None, None, 1.0, 0.0, 0.0)
                Gof_l = ops.linear_dgrad(Ga_l, Lw["wo"], out=new("Gof_l", B, nqd))
                Gho_l, D_l = new("Gho_l", B, nqd), f32("D_l", B, nq, 1)
                ops.attn_bwd_prep(Gof_l, st["o_l"], Gho_l, D_l, B, 1, nq, d, E["pv"], 0.5)
                Gho = zeros("Gho", M, nqd).index_copy_(0, fw["last"], Gho_l)
                D = ar.get("D", (B, nq, S), torch.float32, zero=True)
                D.view(B * nq, S)[:, S - 1].copy_(D_l.view(B * nq))
                Gs1 = zeros("Gs1", M, H).index_copy_(0, fw["last"], Gs1_l)
                q_begin = S - 1
            else:
                # ---- post-feed-forward norm, gated MLP, pre-feed-forward norm + residual
                if Gdn is None:      # (with the site kernels the layer above has produced Gdn together with Gs)
                    Gdn = new("Gdn", M, H)
                    ops.rmsnorm_bwd_add2(None, Gs, Lw["ln_pff"], st["rstd_pff"], None, None, Gdn, None, None, 1.0, 0.0, 0.0)
                if st["coef"]:
                    Agu = ops.gemm_gated_bwd_coef(Gdn, Lw["wd"], st["gu"], wide("Agu", M, 2 * I))
                else:
                    Agu = ops.gemm_gated_bwd(Gdn, Lw["wd"], st["gu"], wide("Agu", M, 2 * I), self.eps_g, E["lin"], self.act)
                Gx2 = ops.linear_dgrad(Agu, Lw["wgu"], out=new("Gx2", M, H))
                Gs1, Ga = new("Gs1", M, H), new("Ga", M, H)
                if site:      # pre-feed-forward norm + residual (gradient w.r.t. h1) and the post-attention norm in one pass
                    ops.sandwich_norm_bwd(Gs, Gx2, Lw["ln_pf"], st["rstd2"], Lw["ln_pa"], st["rstd_pa"], Gs1, Ga, 1.0)
                else:
                    ops.rmsnorm_bwd_add2(Gs, Gx2, Lw["ln_pf"], st["rstd2"], None, None, Gs1, None, None, 1.0, 0.0, 0.0)     # w.r.t. h1
                    # ---- post-attention norm, o-proj, attention
                    ops.rmsnorm_bwd_add2(None, Gs1, Lw["ln_pa"], st["rstd_pa"], None, None, Ga, None, None, 1.0, 0.0, 0.0)
                Gof = ops.linear_dgrad(Ga, Lw["wo"], out=new("Gof", M, nqd))
                Gho, D = new("Gho", M, nqd), f32("D", B, nq, S)
                ops.attn_bwd_prep(Gof, st["o"], Gho, D, B, S, nq, d, E["pv"], 0.5)
            q, k, v = st["qr"], st["kr"], st["qkv"][:, nqd + nkd:]
            k_t = q_t = Gho_t = None
            if self.attn_t:
                k_t = ops.transpose_heads(k, B, S, nk, d)
                q_t = ops.transpose_heads(q, B, S, nq, d)
                Gho_t = ops.transpose_heads(Gho, B, S, nq, d)
            dq = new("dq", M, nqd) if q_begin == 0 else zeros("dq", M, nqd)          # (queries below q_begin carry no relevance: rows stay zero)
            dk_h, dv_h = new("dk_h", M, nqd), new("dv_h", M, nqd)
            ops.attn_bwd_dq(q, k, v, k_t, Gho, st["lse"], D, dq, B, S, nq, nk, d, c["scale"], E["mask"], E["qk"], causal, win, q_begin=q_begin,
                            row_iv=iv)
            ops.attn_bwd_dkv(q, k, v, q_t, Gho, Gho_t, st["lse"], D, dk_h, dv_h, B, S, nq, nk, d, c["scale"], E["mask"], E["qk"], causal, win,
                             q_begin=q_begin, row_iv=iv)
            Aqkv = new("Aqkv", M, nqkv)
            if site:      # the group sums, RoPE's backward and the q / k norms' scale in one pass over dq / dk_h / dv_h
                ops.qkv_bwd_pack(dq, dk_h, dv_h, Lw["qn"], Lw["kn"], st["rstd_q"], st["rstd_k"], cos, sin, Aqkv, S, nq, nk, d, 1.0)
            else:
                dk = ops.gqa_reduce(dk_h, new("dk", M, nkd), M, nk, rep, d)
                ops.gqa_reduce(dv_h, Aqkv[:, nqd + nkd:], M, nk, rep, d)
                # RoPE backward (a rotation: plain gradient), then the q / k norms' row-constant scale, written into the fused [q | k | v] operand
                Gqn = ops.rope_bwd(dq, None, None, new("Gqn", M, nqd), cos, sin, S, nq, d, 0.0, 0.0)
                Gkn = ops.rope_bwd(dk, None, None, new("Gkn", M, nkd), cos, sin, S, nk, d, 0.0, 0.0)
                ops.head_rmsnorm_bwd(Gqn, Lw["qn"], st["rstd_q"], Aqkv[:, :nqd], nq, d, 1.0)
                ops.head_rmsnorm_bwd(Gkn, Lw["kn"], st["rstd_k"], Aqkv[:, nqd: nqd + nkd], nk, d, 1.0)
            Gx = ops.linear_dgrad(Aqkv, Lw["wqkv"], out=new("Gx", M, H))
            # ---- input norm + residual (and, with the site kernels, the post-feed-forward norm of the layer below)
            Gs = new(("Gs", li & 1), M, H)
            if site and li > 0:
                below = fw["stash"][li - 1]
                Gdn = new("Gdn", M, H)
                ops.sandwich_norm_bwd(Gs1, Gx, Lw["ln_in"], st["rstd1"], self.layers[li - 1]["ln_pff"], below["rstd_pff"], Gs, Gdn, 1.0)
            else:
                Gdn = None
                ops.rmsnorm_bwd_add2(Gs1, Gx, Lw["ln_in"], st["rstd1"], None, None, Gs, None, None, 1.0, 0.0, 0.0)
        return Gs

    # ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def explain(self, input_ids=None, inputs_embeds=None, target=None, return_G=False, lengths=None):
        """input_ids [B, S] (or inputs_embeds [B, S, H] = HF's scaled embeddings); target: None (arg-max of the last position) or [B]
        vocabulary indices.  Returns dict(idx [B], logit [B], R_tok [B, S] fp32 = sum_h e (*) dlogit/de, logits [B, V]).
        lengths [B] (optional): prompts of different lengths in one call, LEFT-padded to S (as LlamaLRP.explain: pad keys are masked through
        the attention kernels' per-row key intervals, RoPE is relative, R_tok is exactly 0 at pad positions)."""
        if inputs_embeds is None:
            input_ids = input_ids.to(self.device)
            B, S = input_ids.shape
            emb = self.embed.index_select(0, input_ids.reshape(-1)) * self.embed_scale.to(self.device)
        else:
            B, S = inputs_embeds.shape[:2]
            emb = inputs_embeds.to(device=self.device, dtype=self.dtype).reshape(B * S, -1).contiguous()
        if S > self.max_seq:
            raise ValueError(f"sequence length {S} exceeds max_seq={self.max_seq}")
        row_iv = None
        if lengths is not None:
            lens = torch.as_tensor(lengths, device=self.device).to(torch.int32).reshape(B)
            if int(lens.min()) < 1 or int(lens.max()) > S:
                raise ValueError("lengths must lie in [1, S]")
            i = torch.arange(S, device=self.device, dtype=torch.int32)
            first = (S - lens)[:, None]
            lo = first.expand(B, S).contiguous()
            hi = torch.where(i[None] >= first, (i + 1)[None].expand(B, S), torch.zeros_like(lo)).contiguous()      # pad rows: empty interval
            row_iv = (lo, hi)
        fw = self.forward(emb, B, S, row_iv)
        if target is None:
            idx, _ = ops.argmax_rows(fw["logits"])
        else:
            tgt = torch.as_tensor(target).reshape(-1).cpu().long()
            if tgt.numel() != B or int(tgt.min()) < 0 or int(tgt.max()) >= self.cfg["vocab"]:
                raise ValueError(f"target must hold {B} vocabulary indices in [0, {self.cfg['vocab']})")
            idx = tgt.to(device=self.device, dtype=torch.int32).contiguous()
        G = self.backward(fw, idx, B, S)
        logits = fw["logits"].clone()
        out = dict(idx=idx, logit=logits.gather(1, idx.long()[:, None])[:, 0], R_tok=ops.readout(emb, G).view(B, S), logits=logits)
        if return_G:
            out["G_emb"], out["emb"] = G.view(B, S, -1).clone(), emb.view(B, S, -1)
        return out
