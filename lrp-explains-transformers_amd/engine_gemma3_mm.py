"""Fused AttnLRP driver for Gemma-3 WITH the image branch (BASELINE config 4: "Gemma-3-4B-it image+text -- ViT patch + text token
relevance"): SigLIP vision tower + Gemma3MultiModalProjector + text decoder as straight sequences of liblrp_hip
launches -- no autograd, no module hooks.  Relevance of the text tokens AND of the pixels / ViT patches for one explained logit.

Semantics = what `lxt.efficient.monkey_patch(modeling_gemma3)` does to a Gemma3ForConditionalGeneration (pinned by the reference
fixture tests/golden/gemma3_mm.npz, both attention implementations):
  * text decoder: lxt/efficient/models/gemma3.py:11-19 as in engine_gemma3.Gemma3LRP; image tokens replace the word embeddings at the
    `image_token_id` positions (HF Gemma3Model.forward: masked_scatter) and attend to each other in BOTH directions (HF
    `create_masks_for_vision_model`): per-row key intervals, causal = 0 (mm_row_intervals below);
  * projector: AvgPool2d over the patch grid (linear: plain gradient), Gemma3RMSNorm -- the gemma3 map patches that CLASS, so the
    projector's norm takes the identity rule too --, projection matmul (plain Linear);
  * SigLIP tower: the gemma3 map patches NOTHING in modeling_siglip: LayerNorm, GELU-tanh and the Linears keep ORDINARY gradients
    (ops.layernorm_bwd_plain, ops.act_grad).  Its attention is looked up in the process-wide attention registry, which monkey_patch
    wraps: with attn_implementation "sdpa" (HF's default) the tower's attention takes the AttnLRP rule (q, k / 4, v / 2:
    lxt/efficient/patches.py:193-203), with "eager" it does not -- `vision_attn_rule` selects which (default True = sdpa).
  * patch embedding: Conv2d with stride = kernel as ONE GEMM over the unfolded patches; pixel relevance = pixel (*) d logit / d pixel,
    patch relevance = its sum over a patch.
MI355X notes: head_dim 72 is zero-padded to 96 (bf16; fp32 parity runs: 128) and the MLP width 4304 to 4352 inside the fused weights (exact: padded rows / columns
are zero), so every contraction runs on the 256 x 256 MFMA GEMM and the 32 x 32 attention kernels; the average pool is a GEMM with a
constant [tokens, patches] matrix (power-of-two pooling windows: 1 / k^2 exact in bf16)."""
import torch

from . import ops
from .engine_gemma3 import Gemma3LRP


def mm_row_intervals(token_type_ids, window):
    """token_type_ids [B, S] (1 = image token) -> {"global": (lo, hi), "local": (lo, hi)} int32 [B, S] per-row key intervals [lo, hi):
    causal for text rows, the whole image block (contiguous run of image tokens) for image rows; "local" adds the sliding window
    (key > query - window) -- the union stays an interval because a block contains its own rows.  HF: create_masks_for_vision_model
    (causal / sliding mask OR same-image-block)."""
    tt = torch.as_tensor(token_type_ids).cpu().bool()
    B, S = tt.shape
    i = torch.arange(S)
    start, end = torch.zeros(B, S, dtype=torch.long), torch.zeros(B, S, dtype=torch.long)
    for b in range(B):
        row = tt[b]
        prev = torch.cat([torch.tensor([False]), row[:-1]])
        nxt = torch.cat([row[1:], torch.tensor([False])])
        s_idx, e_idx = (row & ~prev).nonzero()[:, 0], (row & ~nxt).nonzero()[:, 0] + 1
        blk = torch.cumsum((row & ~prev).long(), 0) - 1
        start[b] = torch.where(row, s_idx[blk.clamp_min(0)] if len(s_idx) else i, i)
        end[b] = torch.where(row, e_idx[blk.clamp_min(0)] if len(e_idx) else i + 1, i + 1)
    lo_g = torch.zeros(B, S, dtype=torch.long)
    lo_l = torch.minimum((i - window + 1).clamp_min(0)[None].expand(B, S), torch.where(tt, start, i[None].expand(B, S)))
    f = lambda x: x.to(torch.int32).contiguous()       # noqa: E731
    return {"global": (f(lo_g), f(end)), "local": (f(lo_l), f(end))}


def vision_weights_from_hf(model):
    """plain tensors of the SigLIP tower + projector of a HF Gemma3ForConditionalGeneration (no copies)"""
    core = model.model if hasattr(model, "lm_head") else model
    vt = core.vision_tower.vision_model if hasattr(core.vision_tower, "vision_model") else core.vision_tower
    pj = core.multi_modal_projector
    vc = model.config.vision_config
    d = lambda t: t.detach()        # noqa: E731
    emb = vt.embeddings
    W = dict(patch_w=d(emb.patch_embedding.weight), patch_b=d(emb.patch_embedding.bias), pos=d(emb.position_embedding.weight),
             post_w=d(vt.post_layernorm.weight), post_b=d(vt.post_layernorm.bias), proj_norm=d(pj.mm_soft_emb_norm.weight),
             proj_w=d(pj.mm_input_projection_weight), layers=[])
    for L in vt.encoder.layers:
        a, m = L.self_attn, L.mlp
        W["layers"].append(dict(ln1_w=d(L.layer_norm1.weight), ln1_b=d(L.layer_norm1.bias), ln2_w=d(L.layer_norm2.weight), ln2_b=d(L.layer_norm2.bias),
                                wq=d(a.q_proj.weight), bq=d(a.q_proj.bias), wk=d(a.k_proj.weight), bk=d(a.k_proj.bias), wv=d(a.v_proj.weight),
                                bv=d(a.v_proj.bias), wo=d(a.out_proj.weight), bo=d(a.out_proj.bias), w1=d(m.fc1.weight), b1=d(m.fc1.bias),
                                w2=d(m.fc2.weight), b2=d(m.fc2.bias)))
    act = getattr(vc, "hidden_act", "gelu_pytorch_tanh")
    if act not in ("gelu_pytorch_tanh", "gelu_tanh", "gelu"):
        raise NotImplementedError(f"SiglipLRP: activation {act!r}")
    cfg = dict(hidden=vc.hidden_size, inter=vc.intermediate_size, n_layers=vc.num_hidden_layers, n_heads=vc.num_attention_heads,
               image=vc.image_size, patch=vc.patch_size, channels=vc.num_channels, ln_eps=float(vc.layer_norm_eps),
               act="gelu" if act == "gelu" else "gelu_tanh", tokens_per_image=int(model.config.mm_tokens_per_image),
               text_hidden=model.config.text_config.hidden_size, image_token_id=int(model.config.image_token_id))
    return cfg, W


class SiglipLRP:
    """SigLIP tower + Gemma-3 projector: forward(pixel_values) -> image features [n_img * T, H_text]; backward(G_features) -> d / d pixels"""

    def __init__(self, cfg, W, dtype=torch.bfloat16, device="cuda", vision_attn_rule=True):
        if not torch.cuda.is_available():
            raise RuntimeError("SiglipLRP needs a HIP device: the LRP kernels have no CPU fallback")
        self.cfg, self.dtype, self.device, self.attn_rule = dict(cfg), dtype, torch.device(device), bool(vision_attn_rule)
        H, I, nh = cfg["hidden"], cfg["inter"], cfg["n_heads"]
        d0 = H // nh
        # head dim the attention kernels are built for (SigLIP: 72 -> 96 in bf16: the 32 x 32 kernels exist for 64 / 96 / 128 / 256; fp32: -> 128)
        dp = next(c for c in ((32, 64, 96, 128, 256) if dtype == torch.bfloat16 else (32, 64, 128, 256)) if c >= d0)
        Ip = (I + 63) // 64 * 64                                        # MLP width on the 64-element K tiles of the MFMA GEMM (4304 -> 4352)
        self.d0, self.dp, self.Ip = d0, dp, Ip
        g = cfg["image"] // cfg["patch"]
        self.grid, self.P = g, g * g
        T = cfg["tokens_per_image"]
        side = int(round(T ** 0.5))
        if side * side != T or g % side:
            raise NotImplementedError(f"SiglipLRP: {T} tokens per image on a {g} x {g} patch grid")
        self.T, self.k = T, g // side
        t = lambda x: x.to(device=self.device, dtype=dtype)             # noqa: E731
        z = lambda *s: torch.zeros(*s, device=self.device, dtype=dtype)  # noqa: E731
        Cp = cfg["channels"] * cfg["patch"] ** 2
        self.Cp, self.Kp = Cp, (Cp + 7) // 8 * 8
        self.patch_w = z(H, self.Kp)
        self.patch_w[:, :Cp] = t(W["patch_w"]).reshape(H, Cp)
        self.patch_b, self.pos = t(W["patch_b"]).contiguous(), t(W["pos"]).contiguous()
        self.post_w, self.post_b = t(W["post_w"]).contiguous(), t(W["post_b"]).contiguous()
        self.proj_norm = t(W["proj_norm"]).contiguous()
        self.proj_w = t(W["proj_w"]).contiguous()                        # [H_vision, H_text]: the backward's NT operand
        self.proj_wT = ops.transpose(self.proj_w)                        # [H_text, H_vision]: the forward's
        self.layers = []

        def heads_rows(w):                                               # [nh * d0, H] -> [nh * dp, H], zero rows for the padded head dims
            out = z(nh * dp, w.shape[1])
            out.view(nh, dp, -1)[:, :d0] = t(w).view(nh, d0, -1)
            return out

        def heads_vec(b):
            out = z(nh * dp)
            out.view(nh, dp)[:, :d0] = t(b).view(nh, d0)
            return out
        for L in W["layers"]:
            wo = z(H, nh * dp)
            wo.view(H, nh, dp)[:, :, :d0] = t(L["wo"]).view(H, nh, d0)
            w1, b1, w2 = z(Ip, H), z(Ip), z(H, Ip)
            w1[:I], b1[:I], w2[:, :I] = t(L["w1"]), t(L["b1"]), t(L["w2"])
            self.layers.append(dict(
                ln1_w=t(L["ln1_w"]).contiguous(), ln1_b=t(L["ln1_b"]).contiguous(), ln2_w=t(L["ln2_w"]).contiguous(), ln2_b=t(L["ln2_b"]).contiguous(),
                wqkv=torch.cat([heads_rows(L["wq"]), heads_rows(L["wk"]), heads_rows(L["wv"])], 0),
                bqkv=torch.cat([heads_vec(L["bq"]), heads_vec(L["bk"]), heads_vec(L["bv"])], 0), wo=wo, bo=t(L["bo"]).contiguous(),
                w1=w1, b1=b1, w2=w2, b2=t(L["b2"]).contiguous()))
        # average pool over k x k patches as a GEMM with a constant matrix: pooled = A x, A [T, P] (1 / k^2 where patch p lies in token t's window)
        k, side_ = self.k, side
        pr = torch.arange(self.P)
        tok = (pr // g // k) * side_ + (pr % g) // k
        A = torch.zeros(T, self.P)
        A[tok, pr] = 1.0 / (k * k)
        self.pool = A.to(device=self.device, dtype=dtype).contiguous()
        self.poolT = self.pool.t().contiguous()
        self.attn_t = ops.attn_needs_transposed(self.pos, dp)
        self.scale = d0 ** -0.5
        torch.cuda.synchronize(self.device)

    @classmethod
    def from_hf(cls, model, **kw):
        cfg, W = vision_weights_from_hf(model)
        kw.setdefault("dtype", next(model.parameters()).dtype)
        return cls(cfg, W, **kw)

    # ---- pooling over the patch grid of every image: x [n * P, H] <-> [n * T, H]
    def _pool(self, x, n, A):
        H = x.shape[1]
        rows_in, rows_out = A.shape[1], A.shape[0]
        xt = ops.transpose(x.view(n, rows_in, H))                         # [n, H, rows_in]
        yt = ops.gemm_nt(xt, A)                                           # [n, H, rows_out]
        return ops.transpose(yt).reshape(n * rows_out, H)

    def forward(self, pixel_values):
        c = self.cfg
        n, C, Hh, Ww = pixel_values.shape
        kh = c["patch"]
        g, P, H, nh, dp = self.grid, self.P, c["hidden"], c["n_heads"], self.dp
        if Hh // kh != g or Ww // kh != g or C != c["channels"]:
            raise ValueError(f"pixel_values {tuple(pixel_values.shape)}: the tower is built for {c['channels']} x {c['image']} x {c['image']} images")
        pv = pixel_values.to(device=self.device, dtype=self.dtype)
        M = n * P
        patches = torch.zeros(M, self.Kp, device=self.device, dtype=self.dtype)
        patches[:, : self.Cp] = pv[:, :, : g * kh, : g * kh].reshape(n, C, g, kh, g, kh).permute(0, 2, 4, 1, 3, 5).reshape(M, self.Cp)
        h = ops.linear_fwd(patches, self.patch_w, self.patch_b)
        h = ops.add_bcast(h, self.pos)                                     # position embedding: row m gets pos[m % P]
        stash = []
        nhd = nh * dp
        for Lw in self.layers:
            st = dict(h=h)
            x, st["mean1"], st["rstd1"] = ops.layernorm_fwd(h, Lw["ln1_w"], Lw["ln1_b"], c["ln_eps"])
            qkv = ops.linear_fwd(x, Lw["wqkv"], Lw["bqkv"])
            q, k, v = qkv[:, :nhd], qkv[:, nhd: 2 * nhd], qkv[:, 2 * nhd:]
            v_t = ops.transpose_heads(v, n, P, nh, dp) if self.attn_t else None
            o = torch.empty(M, nhd, device=self.device, dtype=self.dtype)
            lse = torch.empty(n, nh, P, device=self.device, dtype=torch.float32)
            ops.attn_fwd(q, k, v, v_t, o, lse, n, P, nh, nh, dp, self.scale, False, 0)
            a = ops.linear_fwd(o, Lw["wo"], Lw["bo"])
            h1 = ops.add_bcast(h, a)
            x2, st["mean2"], st["rstd2"] = ops.layernorm_fwd(h1, Lw["ln2_w"], Lw["ln2_b"], c["ln_eps"])
            z1 = ops.linear_fwd(x2, Lw["w1"], Lw["b1"])
            z2 = ops.linear_fwd(ops.act_fwd(z1, c["act"]), Lw["w2"], Lw["b2"])
            st.update(qkv=qkv, o=o, lse=lse, h1=h1, z1=z1)
            stash.append(st)
            h = ops.add_bcast(h1, z2)
        y, meanf, rstdf = ops.layernorm_fwd(h, self.post_w, self.post_b, c["ln_eps"])
        pooled = self._pool(y, n, self.pool)                                # [n * T, H]
        nrm, rstdp = ops.add_rmsnorm_fwd(pooled, None, self.proj_norm, c["ln_eps"], 1.0)
        feat = ops.linear_fwd(nrm, self.proj_wT)                           # [n * T, H_text]
        return dict(feat=feat, stash=stash, hL=h, meanf=meanf, rstdf=rstdf, rstdp=rstdp, n=n, pv=pv, patches=patches)

    def backward(self, fw, G_feat):
        """G_feat [n * T, H_text] = d logit / d image features -> (d logit / d pixel_values [n, C, Hh, Ww], the same per unfolded patch [n P, Kp])"""
        c = self.cfg
        n, P, H, nh, dp = fw["n"], self.P, c["hidden"], c["n_heads"], self.dp
        M, nhd = n * P, nh * dp
        Gn = ops.linear_dgrad(G_feat.contiguous(), self.proj_wT)            # c = s W with W = proj^T [H_text, H_vision]
        Gp = torch.empty_like(Gn)
        ops.rmsnorm_bwd_add2(None, Gn, self.proj_norm, fw["rstdp"], None, None, Gp, None, None, 1.0, 0.0, 0.0)    # Gemma3RMSNorm: identity rule
        Gy = self._pool(Gp, n, self.poolT)                                  # [M, H]
        Gh = ops.layernorm_bwd_plain(Gy, fw["hL"], self.post_w, fw["meanf"], fw["rstdf"])
        for Lw, st in zip(reversed(self.layers), reversed(fw["stash"])):
            # MLP branch (plain gradients): fc2 -> GELU' -> fc1 -> LayerNorm VJP, added to the residual gradient
            Gm = ops.linear_dgrad(Gh, Lw["w2"])
            Gz1 = ops.act_grad(Gm, st["z1"], c["act"])
            Gx2 = ops.linear_dgrad(Gz1, Lw["w1"])
            Gh1 = ops.add_bcast(Gh, ops.layernorm_bwd_plain(Gx2, st["h1"], Lw["ln2_w"], st["mean2"], st["rstd2"]))
            # attention branch
            Go = ops.linear_dgrad(Gh1, Lw["wo"])
            qkv = st["qkv"]
            q, k, v = qkv[:, :nhd], qkv[:, nhd: 2 * nhd], qkv[:, 2 * nhd:]
            Gho, D = torch.empty_like(Go), torch.empty(n, nh, P, device=self.device, dtype=torch.float32)
            ops.attn_bwd_prep(Go, st["o"], Gho, D, n, P, nh, dp, 0.0, 0.5 if self.attn_rule else 1.0)
            k_t = q_t = Gho_t = None
            if self.attn_t:
                k_t, q_t = ops.transpose_heads(k, n, P, nh, dp), ops.transpose_heads(q, n, P, nh, dp)
                Gho_t = ops.transpose_heads(Gho, n, P, nh, dp)
            Aqkv = torch.empty_like(qkv)
            dq, dk, dv = Aqkv[:, :nhd], Aqkv[:, nhd: 2 * nhd], Aqkv[:, 2 * nhd:]
            ops.attn_bwd_dq(q, k, v, k_t, Gho, st["lse"], D, dq, n, P, nh, nh, dp, self.scale, 0.0, 0.0, False, 0)
            ops.attn_bwd_dkv(q, k, v, q_t, Gho, Gho_t, st["lse"], D, dk, dv, n, P, nh, nh, dp, self.scale, 0.0, 0.0, False, 0)
            if not self.attn_rule:
                # the kernels carry the uniform rule's 1/2 on the QK^T side (dq, dk = 1/2 scale dS K): the un-patched tower wants the plain
                # gradient -- an exact doubling (g z / (c z) with c = 1/2)
                ops.eps_scale2d(dq, dq, dq, 0.5, 0.0)
                ops.eps_scale2d(dk, dk, dk, 0.5, 0.0)
            Gx = ops.linear_dgrad(Aqkv, Lw["wqkv"])
            Gh = ops.add_bcast(Gh1, ops.layernorm_bwd_plain(Gx, st["h"], Lw["ln1_w"], st["mean1"], st["rstd1"]))
        # the position embedding is an added constant; patch embedding = one GEMM over the unfolded patches
        Gpatch = ops.linear_dgrad(Gh, self.patch_w)                         # [M, Kp]
        kh, g, C = c["patch"], self.grid, c["channels"]
        Gpix = torch.zeros_like(fw["pv"])
        Gpix[:, :, : g * kh, : g * kh] = Gpatch[:, : self.Cp].reshape(n, g, g, C, kh, kh).permute(0, 3, 1, 4, 2, 5).reshape(n, C, g * kh, g * kh)
        return Gpix, Gpatch


class Gemma3MMLRP:
    """explain(input_ids, pixel_values) -> dict(idx, logit, R_tok [B, S], R_pix [n_img, C, H, W], R_patch [n_img, g, g], logits):
    AttnLRP (lxt.efficient) relevance of text tokens and image pixels / ViT patches of a Gemma-3 image + text prompt"""

    def __init__(self, text, vision):
        self.text, self.vision = text, vision
        self.image_token_id = vision.cfg["image_token_id"]

    @classmethod
    def from_hf(cls, model, dtype=None, device="cuda", max_seq=4096, vision_attn_rule=True):
        dtype = dtype or next(model.parameters()).dtype
        return cls(Gemma3LRP.from_hf(model, dtype=dtype, device=device, max_seq=max_seq),
                   SiglipLRP.from_hf(model, dtype=dtype, device=device, vision_attn_rule=vision_attn_rule))

    @torch.no_grad()
    def explain(self, input_ids, pixel_values, token_type_ids=None, target=None, attention_mask=None, lengths=None):
        """Prompts of ONE length only: this driver builds its masks from the image-token positions alone (causal text, bidirectional image blocks),
        so a padded batch would attend to its pad tokens -- refused loudly (use one call per length, or the drop-in path
        lxt_amd.efficient.monkey_patch(modeling_gemma3), which takes HF's padding masks).  Also unlike LlamaLRP / Gemma3LRP: no workspace arena and
        no hipGraph capture yet -- the tower allocates its stashes per call (visible as host_issue_frac in bench.py's config4 image+text line)."""
        tx, vi = self.text, self.vision
        dev = tx.device
        ids = input_ids.to(dev)
        B, S = ids.shape
        if lengths is not None and any(int(n) != S for n in lengths):
            raise NotImplementedError("Gemma3MMLRP.explain: padded image + text batches are not supported (every prompt must fill the batch's sequence length)")
        if attention_mask is not None and not bool(torch.as_tensor(attention_mask).bool().all()):
            raise NotImplementedError("Gemma3MMLRP.explain: padded image + text batches are not supported (attention_mask has masked positions)")
        # the prompt's bookkeeping (image positions, per-row key intervals) is HOST work on the host copy of the ids: with ids handed over on the CPU
        # (the usual tokenizer output) explain() contains no device -> host round trip ahead of the launches -- three of them (nonzero, any, the
        # interval builder's .cpu()) used to drain the queue at the top of every call, and the tower's many short kernels then ran behind the host
        ids_h = input_ids if input_ids.device.type == "cpu" else input_ids.cpu()
        is_img_h = ids_h == self.image_token_id
        tt_h = is_img_h if token_type_ids is None else torch.as_tensor(token_type_ids).cpu().bool()
        rows_h = is_img_h.reshape(-1).nonzero()[:, 0]                            # flat positions of the image tokens, in order
        n_img = pixel_values.shape[0]
        if rows_h.numel() != n_img * vi.T:
            raise ValueError(f"{rows_h.numel()} image tokens in input_ids for {n_img} images of {vi.T} tokens each")
        iv_h = mm_row_intervals(tt_h, tx.cfg["window"]) if bool(tt_h.any()) else None
        rows, is_img = rows_h.to(dev), is_img_h.to(dev)
        fv = vi.forward(pixel_values)
        safe = torch.where(is_img, torch.zeros_like(ids), ids) if self.image_token_id >= tx.cfg["vocab"] else ids
        emb = tx.embed.index_select(0, safe.reshape(-1)) * tx.embed_scale.to(dev)
        emb.index_copy_(0, rows, fv["feat"].to(emb.dtype))                       # HF: inputs_embeds.masked_scatter(image mask, image features)
        iv = None if iv_h is None else {k: (lo.to(dev), hi.to(dev)) for k, (lo, hi) in iv_h.items()}
        fw = tx.forward(emb, B, S, iv)
        if target is None:
            idx, _ = ops.argmax_rows(fw["logits"])
        else:
            idx = torch.as_tensor(target).reshape(-1).to(device=dev, dtype=torch.int32).contiguous()
        G = tx.backward(fw, idx, B, S)
        R_tok = ops.readout(emb, G)
        R_tok.index_fill_(0, rows, 0.0)                                           # the word embeddings at image positions were replaced: no relevance
        Gpix, Gpatch = vi.backward(fv, G.index_select(0, rows))
        R_pix = ops.mul(fv["pv"].contiguous(), Gpix.contiguous()).float()
        R_patch = ops.readout(fv["patches"], Gpatch).view(n_img, vi.grid, vi.grid)     # sum over a patch of pixel (*) gradient, one row per patch
        logits = fw["logits"].clone()
        return dict(idx=idx, logit=logits.gather(1, idx.long()[:, None])[:, 0], R_tok=R_tok.view(B, S), R_pix=R_pix, R_patch=R_patch,
                    logits=logits)
