"""Whole-model oracle: the reference's EXPLICIT BERT composite in epsilon-faithful gradient form (BASELINE config 2 in
lxt.explicit semantics).

TEST INFRASTRUCTURE (see oracle/__init__.py) -- CPU only, explicit formulas, no autograd.

What it restates (ref = /root/reference/lxt/explicit/models/bert.py):
  rule map        :60-65    nn.Linear -> EpsilonRule (eps 1e-8), GELUActivation / Tanh -> IdentityRule
  embeddings      :249-253  add2(word, token_type.detach()), add2(., position) (eps 1e-8), LayerNormEpsilon
  attention       :338-373  lf.matmul(q, k^T) [R/(2 s + 1e-8)], mul2 by 1/sqrt(d), add2(., mask) (1e-8), lf.softmax,
                            lf.matmul(p, v)   [R/(2 o + 1e-8)  -- NOT the uniform-eps rule the explicit Llama uses for P.V]
  residuals       :396 ff.  LayerNormEpsilon(add2(dense, input)) after attention and after the MLP
  LayerNormEpsilon          lxt/explicit/modules.py:48-54 -> lf.layer_norm: R_in = x * VJP_{std detached}(R/(y + 1e-6))
                            (lxt/explicit/functional.py:606-635)
Every rule ends in `.mul_(input)`, so G := R / input is defined everywhere and the composite equals an ordinary backward pass
whose incoming gradient is multiplied by z/(c z + eps) at each eps site; that form is spelled out here.
Pinned by tests/golden/make_golden_bert_explicit.py against the reference's own Functions (fixture bert_base_explicit.npz).
"""
import math

import torch
import torch.nn.functional as F

EPS = dict(lin=1e-8, add=1e-8, mm=1e-8, mask=1e-8, ln=1e-6)


def ratio(z, c, eps):
    return z / (c * z + eps)


def _ln(x, w, b, var_eps):
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    std = (var + var_eps).sqrt()
    return (x - mean) / std * w + b, std


def _ln_bwd(Gy, y, w, std):
    """G_x of LayerNormEpsilon: u = G_y * y/(y + eps_ln) * w / std ; G_x = u - mean_row(u)"""
    u = Gy * ratio(y, 1, EPS["ln"]) * w / std
    return u - u.mean(-1, keepdim=True)


def cast(W, dtype):
    out = {k: (v.to(dtype) if torch.is_tensor(v) else v) for k, v in W.items() if k != "layers"}
    out["layers"] = [{k: v.to(dtype) for k, v in L.items()} for L in W["layers"]]
    return out


def forward(W, ids, rnd=None):
    """ids [S] -> cache (one prompt).  rnd: optional hook applied to every stored activation (tests/util.py:
    fp32_conditioning_bert models fp32-sized evaluation noise with it)"""
    rnd = rnd or (lambda x: x)
    S = ids.shape[0]
    H, nh = W["word"].shape[1], W["heads"]
    d = H // nh
    c = {}
    word = W["word"][ids]
    e1 = rnd(word + W["tt"][0][None])
    e2 = rnd(e1 + W["pos"][:S])
    h, std0 = _ln(e2, W["eln_w"], W["eln_b"], W["ln_eps"])
    h = rnd(h)
    c.update(word=word, e1=e1, e2=e2, h0=h, std0=std0)
    layers = []
    scale = 1 / math.sqrt(d)
    for L in W["layers"]:
        lc = dict(h=h)
        q = rnd(F.linear(h, L["wq"], L["bq"])).view(S, nh, d).transpose(0, 1)
        k = rnd(F.linear(h, L["wk"], L["bk"])).view(S, nh, d).transpose(0, 1)
        v = rnd(F.linear(h, L["wv"], L["bv"])).view(S, nh, d).transpose(0, 1)
        s = rnd(q @ k.transpose(-1, -2))
        p = F.softmax(s * scale, dim=-1)
        o = rnd(p @ v)
        of = o.transpose(0, 1).reshape(S, H)
        a = rnd(F.linear(of, L["wo"], L["bo"]))
        r1 = rnd(a + h)
        h1, std1 = _ln(r1, L["ln1_w"], L["ln1_b"], W["ln_eps"])
        h1 = rnd(h1)
        zi = rnd(F.linear(h1, L["wi"], L["bi"]))
        m = F.gelu(zi)
        dn = rnd(F.linear(m, L["wd"], L["bd"]))
        r2 = rnd(dn + h1)
        h2, std2 = _ln(r2, L["ln2_w"], L["ln2_b"], W["ln_eps"])
        h2 = rnd(h2)
        lc.update(q=q, k=k, v=v, s=s, p=p, o=o, of=of, a=a, r1=r1, h1=h1, std1=std1, zi=zi, m=m, dn=dn, r2=r2, h2=h2, std2=std2)
        layers.append(lc)
        h = h2
    zp = F.linear(h[0], W["pool_w"], W["pool_b"])
    pooled = torch.tanh(zp)
    logits = F.linear(pooled, W["cls_w"], W["cls_b"])
    c.update(layers=layers, hL=h, zp=zp, pooled=pooled, logits=logits, scale=scale)
    return c


def backward(W, c, target):
    """gradient-form explicit backward; returns G at the word embeddings [S,H] and the per-layer sum(h * G_h)"""
    S, H = c["hL"].shape
    nh = W["heads"]
    d = H // nh
    scale = c["scale"]
    z = c["logits"][target]
    # classifier eps rule (one output row), tanh identity rule G_x = G_y * tanh(x)/x, pooler eps rule
    Gpooled = ratio(z, 1, EPS["lin"]) * W["cls_w"][target]
    Gzp = Gpooled * torch.where(c["zp"] == 0, torch.zeros_like(c["zp"]), c["pooled"] / c["zp"])
    Gh = torch.zeros_like(c["hL"])
    Gh[0] = (Gzp * ratio(c["zp"], 1, EPS["lin"])) @ W["pool_w"]
    layer_R = [float((c["hL"] * Gh).sum())]
    for L, lc in zip(reversed(W["layers"]), reversed(c["layers"])):
        # h2 = LN(add2(dn, h1))
        Gr2 = _ln_bwd(Gh, lc["h2"], L["ln2_w"], lc["std2"])
        Gs = Gr2 * ratio(lc["r2"], 1, EPS["add"])                       # add2: the same factor to both summands
        Gm = (Gs * ratio(lc["dn"], 1, EPS["lin"])) @ L["wd"]
        Gzi = Gm * torch.where(lc["zi"] == 0, torch.zeros_like(lc["zi"]), lc["m"] / lc["zi"])       # GELU identity rule
        Gh1 = Gs + (Gzi * ratio(lc["zi"], 1, EPS["lin"])) @ L["wi"]
        # h1 = LN(add2(a, h))
        Gr1 = _ln_bwd(Gh1, lc["h1"], L["ln1_w"], lc["std1"])
        Gs1 = Gr1 * ratio(lc["r1"], 1, EPS["add"])
        Gof = (Gs1 * ratio(lc["a"], 1, EPS["lin"])) @ L["wo"]
        Go = Gof.view(S, nh, d).transpose(0, 1)
        Gho = Go * ratio(lc["o"], 2, EPS["mm"])                         # lf.matmul on P.V: R/(2 o + eps)
        dP = Gho @ lc["v"].transpose(-1, -2)
        dV = lc["p"].transpose(-1, -2) @ Gho
        dS3 = lc["p"] * (dP - (dP * lc["p"]).sum(-1, keepdim=True))     # softmax rule == softmax VJP in gradient form
        s2 = lc["s"] * scale
        dS = dS3 * ratio(s2, 1, EPS["mask"]) * scale                    # add2(., mask = 0), mul2 by the constant
        Ghs = dS * ratio(lc["s"], 2, EPS["mm"])                         # lf.matmul on q k^T
        dQ = Ghs @ lc["k"]
        dK = Ghs.transpose(-1, -2) @ lc["q"]
        Aq = (dQ * ratio(lc["q"], 1, EPS["lin"])).transpose(0, 1).reshape(S, H)
        Ak = (dK * ratio(lc["k"], 1, EPS["lin"])).transpose(0, 1).reshape(S, H)
        Av = (dV * ratio(lc["v"], 1, EPS["lin"])).transpose(0, 1).reshape(S, H)
        Gh = Gs1 + Aq @ L["wq"] + Ak @ L["wk"] + Av @ L["wv"]
        layer_R.append(float((lc["h"] * Gh).sum()))
    Ge2 = _ln_bwd(Gh, c["h0"], W["eln_w"], c["std0"])
    Ge1 = Ge2 * ratio(c["e2"], 1, EPS["add"])
    Gword = Ge1 * ratio(c["e1"], 1, EPS["add"])
    return Gword, layer_R[::-1]


def explain(W, ids, target=None, dtype=torch.float64, rnd=None):
    Wd = cast(W, dtype)
    c = forward(Wd, ids, rnd)
    if target is None:
        target = int(c["logits"].argmax())
    G, layer_R = backward(Wd, c, target)
    # explicit protocol seeds the explained logit with ITS VALUE (relevance = logit): G above is per unit seed gradient 1
    R_emb = c["word"] * G
    return dict(idx=target, logit=float(c["logits"][target]), logits=c["logits"], R_tok=R_emb.sum(-1), R_emb=R_emb, layer_R=layer_R)
