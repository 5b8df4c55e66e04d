"""Whole-model oracle: AttnLRP for a Llama-style decoder in epsilon-faithful gradient form.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- CPU only, explicit formulas, no autograd.

What it restates
----------------
The reference's explicit Llama composite (lxt/explicit/models/llama.py:83-93 rule map,
:226-260 RoPE, :273-281 MLP, :379-391 attention, :481-488 residuals) propagates relevance
R with one autograd.Function per op.  Every such rule ends in ``.mul_(input)``
(lxt/explicit/functional.py:362,405-406,450-451; lxt/explicit/rules.py:221,282), so
G := R / input is always defined and the whole composite is equal to an ORDINARY backward
pass whose incoming gradient is multiplied by  z/(c*z+eps)  at every epsilon-rule site
(z = that op's forward output; c = 1 for linear/add/uniform-eps, c = 2 for lf.matmul).
That "gradient form" is what the HIP engine implements, and what this file spells out.

mode="explicit"  -> reference lxt.explicit semantics (eps terms exactly where the
                    reference has them)
mode="efficient" -> reference lxt.efficient semantics (lxt/efficient/patches.py:111-203,
                    lxt/efficient/rules.py:88-127): no eps terms, activation ratio
                    act(x)/(x+1e-10), attention factors 1/4, 1/4, 1/2.

Pinned against the imported reference by tests/golden/make_golden.py (fixtures
tests/golden/llama_*.npz).
"""
import math
import torch
import torch.nn.functional as F

EXPLICIT = dict(lin=1e-8, add=1e-8, qk=1e-8, mask=1e-8, pv=1e-6, rope=1e-8, act=0.0)
EFFICIENT = dict(lin=0.0, add=0.0, qk=0.0, mask=0.0, pv=0.0, rope=0.0, act=1e-10)


def eps_table(mode):
    if mode == "explicit":
        return dict(EXPLICIT)
    if mode == "efficient":
        return dict(EFFICIENT)
    raise ValueError(mode)


def ratio(z, c, eps):
    """z/(c*z+eps); with eps == 0 the reference has no stabiliser at all -> 1/c."""
    if eps == 0.0:
        return torch.full_like(z, 1.0 / c)
    return z / (c * z + eps)


def config_from_hf(hf_cfg):
    hd = getattr(hf_cfg, "head_dim", None) or hf_cfg.hidden_size // hf_cfg.num_attention_heads
    theta = None
    rp = getattr(hf_cfg, "rope_parameters", None)
    if isinstance(rp, dict):
        theta = rp.get("rope_theta")
    if theta is None:
        theta = getattr(hf_cfg, "rope_theta", 10000.0)
    cfg = dict(hidden=hf_cfg.hidden_size, inter=hf_cfg.intermediate_size,
               n_layers=hf_cfg.num_hidden_layers, n_heads=hf_cfg.num_attention_heads,
               n_kv=hf_cfg.num_key_value_heads, head_dim=hd, vocab=hf_cfg.vocab_size,
               rope_theta=float(theta), rms_eps=float(hf_cfg.rms_norm_eps))
    if isinstance(rp, dict) and rp.get("rope_type", "default") != "default":
        cfg["rope_scaling"] = {k: v for k, v in rp.items() if k != "rope_theta"}
    return cfg


def weights_from_hf(model, dtype=torch.float32):
    """Pull plain tensors out of a HF LlamaForCausalLM (host-side plumbing for the tests)."""
    m = model.model
    W = dict(embed=m.embed_tokens.weight.detach().to(dtype).clone(),
             norm=m.norm.weight.detach().to(dtype).clone(),
             lm_head=model.lm_head.weight.detach().to(dtype).clone(), layers=[])
    for L in m.layers:
        a, p = L.self_attn, L.mlp
        W["layers"].append(dict(
            ln1=L.input_layernorm.weight.detach().to(dtype).clone(),
            ln2=L.post_attention_layernorm.weight.detach().to(dtype).clone(),
            wq=a.q_proj.weight.detach().to(dtype).clone(), wk=a.k_proj.weight.detach().to(dtype).clone(),
            wv=a.v_proj.weight.detach().to(dtype).clone(), wo=a.o_proj.weight.detach().to(dtype).clone(),
            wg=p.gate_proj.weight.detach().to(dtype).clone(), wu=p.up_proj.weight.detach().to(dtype).clone(),
            wd=p.down_proj.weight.detach().to(dtype).clone()))
    return W


def random_weights(cfg, seed=0, dtype=torch.float32, std=0.02):
    """Synthetic N(0, std) weights of a Llama shape (HF default init is N(0, 0.02))."""
    g = torch.Generator().manual_seed(seed)
    H, I, d = cfg["hidden"], cfg["inter"], cfg["head_dim"]
    nq, nk, V = cfg["n_heads"], cfg["n_kv"], cfg["vocab"]

    def rn(*s):
        return (torch.randn(*s, generator=g, dtype=torch.float32) * std).to(dtype)

    W = dict(embed=rn(V, H), norm=torch.ones(H, dtype=dtype), lm_head=rn(V, H), layers=[])
    for _ in range(cfg["n_layers"]):
        W["layers"].append(dict(ln1=torch.ones(H, dtype=dtype), ln2=torch.ones(H, dtype=dtype),
                                wq=rn(nq * d, H), wk=rn(nk * d, H), wv=rn(nk * d, H), wo=rn(H, nq * d),
                                wg=rn(I, H), wu=rn(I, H), wd=rn(H, I)))
    return W


def rope_inv_freq(cfg):
    """-> (inv_freq [d/2] fp32, attention_scaling).  Restates HF's rope initialisers for the static rope types
    (HF:modeling_rope_utils.py `_compute_default/linear_scaling/llama3_parameters`): "default", "linear" (inv_freq / factor),
    "llama3" (Llama-3.1/3.2: low frequencies divided by `factor`, a smooth blend in the medium band).  The reference inherits
    whatever HF computes (lxt patches nothing in the rotary embedding)."""
    d = cfg["head_dim"]
    inv = 1.0 / (cfg["rope_theta"] ** (torch.arange(0, d, 2, dtype=torch.int64).to(torch.float32) / d))
    rs = cfg.get("rope_scaling") or {}
    kind = rs.get("rope_type", "default")
    if kind == "default":
        return inv, 1.0
    if kind == "linear":
        return inv / rs["factor"], 1.0
    if kind == "llama3":
        factor, lo_f, hi_f = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"]
        old_len = rs["original_max_position_embeddings"]
        lo_wl, hi_wl = old_len / lo_f, old_len / hi_f
        wl = 2 * math.pi / inv
        out = torch.where(wl > lo_wl, inv / factor, inv)
        smooth = (old_len / wl - lo_f) / (hi_f - lo_f)
        blended = (1 - smooth) * out / factor + smooth * out
        medium = ~(wl < hi_wl) & ~(wl > lo_wl)
        return torch.where(medium, blended, out), 1.0
    raise NotImplementedError(f"oracle: rope_type {kind!r}")


def rope_tables(cfg, S, dtype):
    """HF LlamaRotaryEmbedding.forward: cos/sin [S, head_dim], fp32 math, then the model dtype."""
    inv, att = rope_inv_freq(cfg)
    pos = torch.arange(S, dtype=torch.float32)
    fr = pos[:, None] * inv[None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return (emb.cos() * att).to(dtype), (emb.sin() * att).to(dtype)


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def rotate_half_T(g):
    """Transpose of rotate_half: y=(-x2, x1) -> g_x1 = g_y2, g_x2 = -g_y1."""
    g1, g2 = g[..., : g.shape[-1] // 2], g[..., g.shape[-1] // 2:]
    return torch.cat((g2, -g1), dim=-1)


def rms(x, w, eps):
    var = x.pow(2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    return w * (x * rstd), rstd


# ----------------------------------------------------------------------------- forward
def _ident(x):
    return x


def round_through(dtype):
    """activation-storage model: every materialised activation is rounded through `dtype` (bf16: what a bf16 run of the
    reference -- or of the HIP engine -- keeps in memory between ops) while the arithmetic stays in the oracle's dtype.
    Used by the BASELINE-size bf16 test to quote the oracle's OWN sensitivity to bf16 storage next to the engine's error."""
    return lambda x: x.to(dtype).to(x.dtype)


def forward(cfg, W, emb, rnd=None, kv_chunk=None):
    """emb [S,H] -> cache of every activation the backward needs (one prompt).  rnd: optional storage-rounding model
    (round_through); attention scores / probabilities stay un-rounded (fused attention keeps them on chip).
    kv_chunk (optional): evaluate the attention of `kv_chunk` kv groups at a time and do NOT keep scores / probabilities ([heads,S,S]:
    4.3 GB each in fp64 at 32 heads, S = 4096); the backward recomputes them per chunk from the cached q / k -- same operations on the
    same data, only the memory high-water mark changes (BASELINE config 5's real head count on a 62-GB host)."""
    R = rnd or _ident
    S, H = emb.shape
    d, nq, nk = cfg["head_dim"], cfg["n_heads"], cfg["n_kv"]
    rep = nq // nk
    cos, sin = rope_tables(cfg, S, emb.dtype)
    scale = d ** -0.5
    causal = torch.ones(S, S, dtype=torch.bool).tril()
    h = emb
    layers = []
    for Lw in W["layers"]:
        c = dict(h=h)
        x, c["rstd1"] = rms(h, Lw["ln1"], cfg["rms_eps"])
        x = R(x)
        c["x"] = x
        q = R(x @ Lw["wq"].T).view(S, nq, d).transpose(0, 1)      # [nq,S,d]
        k = R(x @ Lw["wk"].T).view(S, nk, d).transpose(0, 1)
        v = R(x @ Lw["wv"].T).view(S, nk, d).transpose(0, 1)
        qr = R(q * cos + rotate_half(q) * sin)
        kr = R(k * cos + rotate_half(k) * sin)
        if kv_chunk is None:
            kx = kr.repeat_interleave(rep, dim=0)
            vx = v.repeat_interleave(rep, dim=0)
            s = qr @ kx.transpose(-1, -2)                          # raw scores (lf.matmul output)
            s2 = s * scale
            s3 = s2.masked_fill(~causal, float("-inf"))
            p = F.softmax(s3, dim=-1)
            o = R(p @ vx)                                          # [nq,S,d]
        else:
            o = torch.empty_like(qr)
            for g0 in range(0, nk, kv_chunk):
                g1 = min(nk, g0 + kv_chunk)
                s_c, p_c = _scores_probs(qr[g0 * rep: g1 * rep], kr[g0:g1], rep, scale, causal)
                o[g0 * rep: g1 * rep] = R(p_c @ v[g0:g1].repeat_interleave(rep, dim=0))
                del s_c, p_c
            s = p = None
        of = o.transpose(0, 1).reshape(S, nq * d)
        a = R(of @ Lw["wo"].T)
        h1 = R(h + a)
        x2, c["rstd2"] = rms(h1, Lw["ln2"], cfg["rms_eps"])
        x2 = R(x2)
        g = R(x2 @ Lw["wg"].T)
        u = R(x2 @ Lw["wu"].T)
        act = F.silu(g)
        m = R(act * u)
        dn = R(m @ Lw["wd"].T)
        h2 = R(h1 + dn)
        c.update(q=q, k=k, v=v, qr=qr, kr=kr, s=s, p=p, o=o, of=of, a=a, h1=h1, x2=x2, g=g, u=u,
                 act=act, m=m, dn=dn, h2=h2)
        layers.append(c)
        h = h2
    xn, rstdf = rms(h, W["norm"], cfg["rms_eps"])
    xn = R(xn)
    logits_last = xn[-1] @ W["lm_head"].T
    return dict(layers=layers, hf=h, xn=xn, rstdf=rstdf, logits_last=logits_last, cos=cos, sin=sin,
                scale=scale, causal=causal, kv_chunk=kv_chunk)


def _scores_probs(qr_c, kr_c, rep, scale, causal):
    """raw scores s = q k^T and p = softmax(s * scale + causal mask) of the query heads of a range of kv groups"""
    s = qr_c @ kr_c.repeat_interleave(rep, dim=0).transpose(-1, -2)
    p = F.softmax((s * scale).masked_fill(~causal, float("-inf")), dim=-1)
    return s, p


# ----------------------------------------------------------------------------- backward
def backward(cfg, W, cache, target, mode="explicit", seed=None, rnd=None):
    """Gradient-form LRP backward.  Returns G at the embedding [S,H] and per-layer sum(h*G_h).
    seed [V] (optional, instead of target): what the user hands to `logits[0,-1].backward(seed)` -- a GRADIENT over the
    last-position logits in efficient mode (contrastive explanations, ref docs/source/quickstart.rst:267-270), a
    RELEVANCE over them in explicit mode (ref examples/paper/llama.py:45: `.backward(logit)` is the one-hot case)."""
    E = eps_table(mode)
    R = rnd or _ident
    S = cache["hf"].shape[0]
    d, nq, nk = cfg["head_dim"], cfg["n_heads"], cfg["n_kv"]
    rep = nq // nk
    cos, sin, scale, causal = cache["cos"], cache["sin"], cache["scale"], cache["causal"]
    dt = cache["hf"].dtype

    # seed: explicit .backward(logit) <=> G = 1 at the explained logit
    if seed is None:
        z = cache["logits_last"][target]
        g_xn_last = ratio(z, 1, E["lin"]) * W["lm_head"][target]   # Linear eps rule, single row/col
    else:
        zl = cache["logits_last"]
        coef = seed.to(zl.dtype) / (zl + E["lin"]) if mode == "explicit" else seed.to(zl.dtype)
        g_xn_last = coef @ W["lm_head"]                            # Linear eps rule over every seeded logit
    Gh = torch.zeros(S, cfg["hidden"], dtype=dt)
    Gh[-1] = R(g_xn_last * W["norm"] * cache["rstdf"][-1])         # RMSNorm identity rule
    layer_R = [float((cache["hf"] * Gh).sum())]

    for Lw, c in zip(reversed(W["layers"]), reversed(cache["layers"])):
        # h2 = add2(h1, dn)
        Gs = R(Gh * ratio(c["h2"], 1, E["add"]))
        # down_proj eps rule
        Gm = R(R(Gs * ratio(c["dn"], 1, E["lin"])) @ Lw["wd"])
        # uniform rule on act*u, identity rule on silu
        Gu = 0.5 * Gm * c["act"]
        Gact = 0.5 * Gm * c["u"]
        if mode == "explicit":
            # G_g * g/(g+eps) with G_g = G_act*act/g  ->  G_act*act/(g+eps)   (no 0/0 at g=0)
            Ag = Gact * (c["act"] / (c["g"] + E["lin"]))
        else:
            Ag = Gact * (c["act"] / (c["g"] + E["act"]))           # efficient: act/(g+1e-10), plain Linear
        Au = Gu * ratio(c["u"], 1, E["lin"])
        Gx2 = R(R(Ag) @ Lw["wg"] + R(Au) @ Lw["wu"])
        Gh1 = Gs + Gx2 * Lw["ln2"] * c["rstd2"]
        # h1 = add2(h, a)
        Gs1 = R(Gh1 * ratio(c["h1"], 1, E["add"]))
        Gof = R(R(Gs1 * ratio(c["a"], 1, E["lin"])) @ Lw["wo"])
        Go = Gof.view(S, nq, d).transpose(0, 1)                    # [nq,S,d]
        # P.V : UniformEpsilon rule (c=1, then /2)
        Ghat_o = R(0.5 * Go * ratio(c["o"], 1, E["pv"]))
        def attn_core(h0, h1_, g0, g1, s_c, p_c):
            """softmax rule (Prop 3.1), add2(mask), mul2(scale), lf.matmul, uniform-eps P.V for query heads h0..h1_ (kv groups g0..g1)"""
            vx = c["v"][g0:g1].repeat_interleave(rep, dim=0)
            kx = c["kr"][g0:g1].repeat_interleave(rep, dim=0)
            Gh_c = Ghat_o[h0:h1_]
            dP = Gh_c @ vx.transpose(-1, -2)
            dVx = p_c.transpose(-1, -2) @ Gh_c
            # softmax (Prop 3.1 == ordinary softmax VJP in gradient form)
            dS3 = p_c * (dP - (dP * p_c).sum(-1, keepdim=True))
            # add2(s2, mask): unmasked entries have s3 = s2
            s2 = s_c * scale
            dS2 = torch.where(causal, dS3 * ratio(s2, 1, E["mask"]), torch.zeros_like(dS3))
            dS = dS2 * scale                                        # mul2 by the constant 1/sqrt(d)
            Ghat_s = dS * ratio(s_c, 2, E["qk"])                    # lf.matmul: R/(2 s + eps)
            return R(Ghat_s @ kx), R(Ghat_s.transpose(-1, -2) @ c["qr"][h0:h1_]), dVx

        kvc = cache.get("kv_chunk")
        if kvc is None:
            dQr, dKx, dVx = attn_core(0, nq, 0, nk, c["s"], c["p"])
        else:
            dQr, dKx, dVx = torch.empty_like(c["qr"]), torch.empty_like(c["qr"]), torch.empty_like(c["qr"])
            for g0 in range(0, nk, kvc):
                g1 = min(nk, g0 + kvc)
                s_c, p_c = _scores_probs(c["qr"][g0 * rep: g1 * rep], c["kr"][g0:g1], rep, scale, causal)
                dQr[g0 * rep: g1 * rep], dKx[g0 * rep: g1 * rep], dVx[g0 * rep: g1 * rep] = attn_core(g0 * rep, g1 * rep, g0, g1, s_c, p_c)
                del s_c, p_c
        dKr = R(dKx.view(nk, rep, S, d).sum(1))
        dV = R(R(dVx).view(nk, rep, S, d).sum(1))
        # RoPE: add2 eps on the rotated tensor, then ordinary transpose of the rotation
        def rope_bwd(Gr, r):
            Gp = Gr * ratio(r, 1, E["rope"])
            return Gp * cos + rotate_half_T(Gp * sin)
        Gq = rope_bwd(dQr, c["qr"])
        Gk = rope_bwd(dKr, c["kr"])
        Aq = (Gq * ratio(c["q"], 1, E["lin"])).transpose(0, 1).reshape(S, nq * d)
        Ak = (Gk * ratio(c["k"], 1, E["lin"])).transpose(0, 1).reshape(S, nk * d)
        Av = (dV * ratio(c["v"], 1, E["lin"])).transpose(0, 1).reshape(S, nk * d)
        Gx = R(R(Aq) @ Lw["wq"] + R(Ak) @ Lw["wk"] + R(Av) @ Lw["wv"])
        Gh = R(Gs1 + Gx * Lw["ln1"] * c["rstd1"])
        layer_R.append(float((c["h"] * Gh).sum()))
    return Gh, layer_R[::-1]


def explain(cfg, W, ids=None, emb=None, target=None, mode="explicit", dtype=torch.float32, seed=None, rnd=None, kv_chunk=None):
    """One explanation: returns dict(idx, logit, R_tok [S], R_emb [S,H], layer_R [L+1])."""
    Wd = cast_weights(W, dtype)
    if emb is None:
        emb = Wd["embed"][ids]
    emb = emb.to(dtype)
    cache = forward(cfg, Wd, emb, rnd=rnd, kv_chunk=kv_chunk)
    if target is None:
        target = int(cache["logits_last"].argmax())
    G, layer_R = backward(cfg, Wd, cache, target, mode, seed=seed, rnd=rnd)
    R_emb = emb * G
    return dict(idx=target, logit=float(cache["logits_last"][target]), R_tok=R_emb.sum(-1),
                R_emb=R_emb, layer_R=layer_R, logits_last=cache["logits_last"])


def cast_weights(W, dtype):
    out = {k: (v.to(dtype) if torch.is_tensor(v) else v) for k, v in W.items() if k != "layers"}
    out["layers"] = [{k: v.to(dtype) for k, v in L.items()} for L in W["layers"]]
    return out
