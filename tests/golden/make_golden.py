#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REAL REFERENCE.

Run in the build container only (needs /root/reference, which does not travel):

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_golden.py

It imports the reference's own Functions (lxt.explicit.functional / rules / modules,
lxt.efficient) and records, for seeded inputs, the outputs THEY produce.  The fixtures are
data only (inputs + expected outputs); no reference source is stored.  While generating it
also asserts that the repo's oracle (oracle/) agrees with the reference, which is what pins
the oracle (see oracle/__init__.py).

Whole-model "explicit" reference: lxt.explicit.models.* cannot be imported under
transformers 5.x (SURVEY.md finding 9), so the decoder is hand-composed here from the
importable reference Functions in exactly the order/eps of the reference's Llama composite
(lxt/explicit/models/llama.py:83-93,226-260,273-281,379-391,481-488).
"""
import math
import os
import sys
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")

import lxt.explicit.functional as lf          # noqa: E402  (the reference)
import lxt.explicit.rules as lrules           # noqa: E402
import lxt.explicit.modules as lm             # noqa: E402
import lxt.efficient.rules as erules          # noqa: E402
from oracle import rules as orules            # noqa: E402  (ours)
from oracle import llama as ollama            # noqa: E402


def nmax(a, b):
    """normalised max error  max|a-b| / max|b|"""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(f"  wrote {name}: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in out.items()))


# ============================================================================ rule-level
def golden_rules():
    print("rule-level fixtures")
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    fx = {}

    # ---- a1 linear eps-rule: C1 shape (768->768, M=1) + a toy shape; consistent relevance R=z*g
    for tag, (M, K, N) in dict(c1=(1, 768, 768), toy=(16, 10, 5), mid=(33, 200, 70)).items():
        x = rn(M, K)
        W = rn(N, K) / math.sqrt(K)
        b = rn(N) * 0.02
        gg = rn(M, N)
        for eps_tag, eps in (("f", 1e-6), ("r", 1e-8)):
            xr = x.clone().requires_grad_()
            if eps_tag == "f":
                z = lf.linear_epsilon(xr, W, b, eps)
            else:
                z = lrules.EpsilonRule(partial(F.linear, weight=W, bias=b), eps)(xr)
            R_out = (z.detach() * gg)
            R_in, = torch.autograd.grad(z, xr, R_out)
            z_o, R_o = orules.linear_epsilon(x, W, b, R_out, eps)
            assert torch.equal(z_o, z.detach()) and torch.equal(R_o, R_in), ("linear", tag, eps_tag)
            fx[f"lin_{tag}_{eps_tag}_Rin"] = R_in
        fx[f"lin_{tag}_x"], fx[f"lin_{tag}_W"], fx[f"lin_{tag}_b"], fx[f"lin_{tag}_g"] = x, W, b, gg

    # ---- a2 matmul rule (Prop 3.3), eps 1e-8 wrapper default
    a, b = rn(2, 3, 10, 32), rn(2, 3, 32, 7)
    gg = rn(2, 3, 10, 7)
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    o = lf.matmul(ar, br)
    R_out = o.detach() * gg
    Ra, Rb = torch.autograd.grad(o, (ar, br), R_out)
    o_o, Ra_o, Rb_o = orules.matmul(a, b, R_out, 1e-8)
    assert nmax(Ra_o, Ra) < 1e-6 and nmax(Rb_o, Rb) < 1e-6
    fx.update(mm_a=a, mm_b=b, mm_g=gg, mm_Ra=Ra, mm_Rb=Rb)

    # ---- a3 softmax rule (Prop 3.1) incl. -inf mask entries
    x = rn(2, 4, 12, 12)
    mask = torch.ones(12, 12, dtype=torch.bool).tril()
    xm = x.masked_fill(~mask, float("-inf"))
    gg = rn(2, 4, 12, 12)
    xr = xm.clone().requires_grad_()
    p = lf.softmax(xr, -1, torch.float32, 1.0, False)
    R_out = p.detach() * gg
    Rx, = torch.autograd.grad(p, xr, R_out)
    p_o, Rx_o = orules.softmax(xm, R_out)
    assert nmax(p_o, p) < 1e-7 and nmax(Rx_o, Rx) < 1e-6
    fx.update(sm_x=xm, sm_g=gg, sm_p=p, sm_Rx=Rx)

    # ---- a5 add2
    a, b, gg = rn(4, 10, 32), rn(4, 10, 32), rn(4, 10, 32)
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    o = lf.add2(ar, br)
    R_out = o.detach() * gg
    Ra, Rb = torch.autograd.grad(o, (ar, br), R_out)
    _, Ra_o, Rb_o = orules.add2(a, b, R_out, 1e-8)
    assert nmax(Ra_o, Ra) < 1e-6 and nmax(Rb_o, Rb) < 1e-6
    fx.update(add_a=a, add_b=b, add_g=gg, add_Ra=Ra, add_Rb=Rb)

    # ---- a6 mul2 (both require grad; and constant second operand)
    a, b, R_out = rn(4, 8), rn(4, 8), rn(4, 8)
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    Ra, Rb = torch.autograd.grad(lf.mul2(ar, br), (ar, br), R_out)
    Ra1, = torch.autograd.grad(lf.mul2(ar, b), (ar,), R_out)
    _, Ra_o, Rb_o = orules.mul2(a, b, R_out)
    _, Ra1_o, _ = orules.mul2(a, b, R_out, True, False)
    assert torch.equal(Ra_o, Ra) and torch.equal(Rb_o, Rb) and torch.equal(Ra1_o, Ra1)
    fx.update(mul_a=a, mul_b=b, mul_R=R_out, mul_Ra=Ra, mul_Rb=Rb, mul_Ra_const=Ra1)

    # ---- mean
    a, R_out = rn(2, 8, 32), rn(2, 8, 1)
    ar = a.clone().requires_grad_()
    Rm, = torch.autograd.grad(lf.mean(ar, -1, True, 1e-6), ar, R_out)
    _, Rm_o = orules.mean(a, R_out, -1, True, 1e-6)
    assert nmax(Rm_o, Rm) < 1e-6
    fx.update(mean_a=a, mean_R=R_out, mean_Rin=Rm)

    # ---- a7 rms_norm_identity
    x, w, R_out = rn(3, 5, 64), rn(64), rn(3, 5, 64)
    xr = x.clone().requires_grad_()
    y = lf.rms_norm_identity(xr, w, 1e-5)
    Rn, = torch.autograd.grad(y, xr, R_out)
    y_o, Rn_o = orules.rms_norm_identity(x, w, 1e-5, R_out)
    assert torch.equal(y_o, y.detach()) and torch.equal(Rn_o, Rn)
    fx.update(rms_x=x, rms_w=w, rms_R=R_out, rms_y=y, rms_Rin=Rn)

    # ---- a9 layer_norm
    x, w, b, gg = rn(2, 7, 48), rn(48), rn(48), rn(2, 7, 48)
    xr = x.clone().requires_grad_()
    y = lf.layer_norm(xr, w, b, 1e-12)
    R_out = y.detach() * gg
    Rl, = torch.autograd.grad(y, xr, R_out)
    y_o, Rl_o = orules.layer_norm(x, w, b, 1e-12, R_out, 1e-6)
    assert nmax(y_o, y) < 1e-6 and nmax(Rl_o, Rl) < 1e-5, nmax(Rl_o, Rl)
    fx.update(ln_x=x, ln_w=w, ln_b=b, ln_g=gg, ln_y=y, ln_Rin=Rl)

    # ---- a4 uniform-epsilon P.V
    pr = F.softmax(rn(2, 3, 9, 9), -1)
    v, gg = rn(2, 3, 9, 16), rn(2, 3, 9, 16)

    class AV(nn.Module):
        def forward(self, a_, v_):
            return torch.matmul(a_, v_)

    prr, vr = pr.clone().requires_grad_(), v.clone().requires_grad_()
    o = lrules.UniformEpsilonRule(AV())(prr, vr)
    R_out = o.detach() * gg
    Rp, Rv = torch.autograd.grad(o, (prr, vr), R_out)
    _, Rp_o, Rv_o = orules.uniform_epsilon_matmul(pr, v, R_out, 1e-6)
    assert nmax(Rp_o, Rp) < 1e-6 and nmax(Rv_o, Rv) < 1e-6
    fx.update(pv_p=pr, pv_v=v, pv_g=gg, pv_Rp=Rp, pv_Rv=Rv)

    # ---- a8 identity / uniform rules
    x, R_out = rn(4, 16), rn(4, 16)
    xr = x.clone().requires_grad_()
    y = lrules.IdentityRule(nn.SiLU())(xr)
    Ri, = torch.autograd.grad(y, xr, R_out)
    assert torch.equal(Ri, R_out)

    # ---- a10/a11 efficient primitives (gradient form)
    x, G = rn(6, 40), rn(6, 40)
    xr = x.clone().requires_grad_()
    y = erules.identity_rule_implicit(F.silu, xr)
    Gi, = torch.autograd.grad(y, xr, G)
    y_o, Gi_o = orules.identity_rule_implicit(F.silu, x, G)
    assert torch.equal(y_o, y.detach()) and nmax(Gi_o, Gi) < 1e-7
    xr = x.clone().requires_grad_()
    yg = erules.identity_rule_implicit(partial(F.gelu, approximate="tanh"), xr)
    Gg, = torch.autograd.grad(yg, xr, G)
    xr = x.clone().requires_grad_()
    Gd, = torch.autograd.grad(erules.divide_gradient(xr, 4), xr, G)
    assert torch.equal(Gd, G / 4)
    fx.update(act_x=x, act_G=G, act_silu_y=y, act_silu_Gin=Gi, act_gelut_y=yg, act_gelut_Gin=Gg)

    save("rules.npz", **fx)


# ============================================================================ model-level
class _Mul(nn.Module):
    def forward(self, a, b):
        return a * b


class _AV(nn.Module):
    def forward(self, a, v):
        return torch.matmul(a, v)


def ref_explicit_llama(cfg, W, emb, target=None, seed=None):
    """Hand-composed lxt.explicit Llama (batch 1) from the REFERENCE's Functions.  `seed` [V]: the relevance pattern handed to
    `logits[0, -1].backward(seed)` instead of the explained logit itself (contrastive explanations, ref docs/source/quickstart.rst:267-270)."""
    dt = emb.dtype
    S = emb.shape[0]
    d, nq, nk = cfg["head_dim"], cfg["n_heads"], cfg["n_kv"]
    rep = nq // nk
    cos, sin = ollama.rope_tables(cfg, S, dt)
    cos, sin = cos[None, None], sin[None, None]

    def lin(x, w):                     # nn.Linear -> rules.EpsilonRule (eps 1e-8)  llama.py:90
        return lrules.EpsilonRule(partial(F.linear, weight=w), 1e-8)(x)

    def rot_half(x):                   # llama.py:226-231
        x1, x2 = x[..., : d // 2], x[..., d // 2:]
        return torch.cat((lf.mul2(x2, -1), x1), dim=-1)

    def rope(x):                       # llama.py:258-259
        return lf.add2(lf.mul2(x, cos.detach()), lf.mul2(rot_half(x), sin.detach()))

    def repeat_kv(x):                  # plain expand/reshape
        b, h, s, dd = x.shape
        return x[:, :, None].expand(b, h, rep, s, dd).reshape(b, h * rep, s, dd)

    minv = torch.finfo(dt).min
    mask = torch.full((S, S), minv, dtype=dt).triu(1)[None, None]

    e = emb[None].clone().requires_grad_()
    h = e
    hs = [h]
    for Lw in W["layers"]:
        res = h
        x = lf.rms_norm_identity(h, Lw["ln1"], cfg["rms_eps"])
        q = lin(x, Lw["wq"]).view(1, S, nq, d).transpose(1, 2)
        k = lin(x, Lw["wk"]).view(1, S, nk, d).transpose(1, 2)
        v = lin(x, Lw["wv"]).view(1, S, nk, d).transpose(1, 2)
        q, k = rope(q), rope(k)
        k, v = repeat_kv(k), repeat_kv(v)
        s = lf.mul2(lf.matmul(q, k.transpose(2, 3)), 1 / math.sqrt(d))          # llama.py:379
        s = lf.add2(s, mask)                                                     # llama.py:384
        p = lm.SoftmaxDT(dim=-1)(s.float() if dt != torch.float64 else s).to(dt)  # llama.py:390
        o = lrules.UniformEpsilonRule(_AV(), 1e-6)(p, v)                         # llama.py:391
        o = lin(o.transpose(1, 2).reshape(1, S, nq * d), Lw["wo"])
        h = lf.add2(res, o)                                                      # llama.py:481
        res = h
        x = lf.rms_norm_identity(h, Lw["ln2"], cfg["rms_eps"])
        g = lrules.IdentityRule(nn.SiLU())(lin(x, Lw["wg"]))                     # llama.py:84
        u = lin(x, Lw["wu"])
        dn = lin(lrules.UniformRule(_Mul())(g, u), Lw["wd"])                     # llama.py:86,281
        h = lf.add2(res, dn)                                                     # llama.py:488
        hs.append(h)
    for t in hs[1:]:
        t.retain_grad()
    logits = lin(lf.rms_norm_identity(h, W["norm"], cfg["rms_eps"]), W["lm_head"])
    last = logits[0, -1]
    if target is None:
        target = int(last.argmax())
    if seed is not None:
        last.backward(seed.to(last.dtype))
    else:
        last[target].backward(last[target].detach())                             # examples/paper/llama.py:45
    R_emb = e.grad[0]
    layer_R = [float(R_emb.sum())] + [float(t.grad.sum()) for t in hs[1:]]
    return dict(idx=target, logit=float(last[target]), R_tok=R_emb.sum(-1), R_emb=R_emb, layer_R=layer_R,
                logits_last=last.detach())


def ref_efficient_llama(cfg, W, ids, attn_impl="eager"):
    """The reference's real lxt.efficient path on a HF LlamaForCausalLM carrying our weights."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.models.llama import modeling_llama
    from lxt.efficient import monkey_patch
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkey_patch(modeling_llama)
    hc = LlamaConfig(hidden_size=cfg["hidden"], intermediate_size=cfg["inter"], num_hidden_layers=cfg["n_layers"],
                     num_attention_heads=cfg["n_heads"], num_key_value_heads=cfg["n_kv"], head_dim=cfg["head_dim"],
                     vocab_size=cfg["vocab"], rms_norm_eps=cfg["rms_eps"], max_position_embeddings=4096,
                     rope_parameters=dict(rope_type="default", rope_theta=cfg["rope_theta"]),
                     tie_word_embeddings=False, attn_implementation=attn_impl)
    model = LlamaForCausalLM(hc).eval()
    with torch.no_grad():
        model.model.embed_tokens.weight.copy_(W["embed"])
        model.model.norm.weight.copy_(W["norm"])
        model.lm_head.weight.copy_(W["lm_head"])
        for L, Lw in zip(model.model.layers, W["layers"]):
            L.input_layernorm.weight.copy_(Lw["ln1"]); L.post_attention_layernorm.weight.copy_(Lw["ln2"])
            L.self_attn.q_proj.weight.copy_(Lw["wq"]); L.self_attn.k_proj.weight.copy_(Lw["wk"])
            L.self_attn.v_proj.weight.copy_(Lw["wv"]); L.self_attn.o_proj.weight.copy_(Lw["wo"])
            L.mlp.gate_proj.weight.copy_(Lw["wg"]); L.mlp.up_proj.weight.copy_(Lw["wu"])
            L.mlp.down_proj.weight.copy_(Lw["wd"])
    for p in model.parameters():
        p.requires_grad_(False)
    e = model.get_input_embeddings()(ids[None]).requires_grad_()
    logits = model(inputs_embeds=e, use_cache=False).logits
    last = logits[0, -1]
    idx = int(last.argmax())
    last[idx].backward()
    R_emb = (e * e.grad)[0]
    return dict(idx=idx, logit=float(last[idx]), R_tok=R_emb.float().sum(-1), R_emb=R_emb.detach(),
                logits_last=last.detach())


def wsum(W):
    """weight checksum so a fixture can verify the regenerated synthetic weights"""
    tot = float(W["embed"].double().abs().sum() + W["lm_head"].double().abs().sum())
    for L in W["layers"]:
        tot += sum(float(v.double().abs().sum()) for v in L.values())
    return tot


def golden_models():
    print("model-level fixtures")
    cases = dict(
        tiny=dict(cfg=dict(hidden=64, inter=128, n_layers=2, n_heads=4, n_kv=2, head_dim=16, vocab=96,
                           rope_theta=10000.0, rms_eps=1e-5), S=16, wseed=1, iseed=11),
        mid=dict(cfg=dict(hidden=256, inter=512, n_layers=4, n_heads=8, n_kv=2, head_dim=32, vocab=512,
                          rope_theta=500000.0, rms_eps=1e-5), S=128, wseed=2, iseed=12),
        d128=dict(cfg=dict(hidden=512, inter=1024, n_layers=2, n_heads=4, n_kv=1, head_dim=128, vocab=256,
                           rope_theta=500000.0, rms_eps=1e-5), S=192, wseed=3, iseed=13),
    )
    for name, c in cases.items():
        cfg, S = c["cfg"], c["S"]
        ids = torch.randint(0, cfg["vocab"], (S,), generator=torch.Generator().manual_seed(c["iseed"]))
        # The explicit composite is CHAOTIC wherever an activation z lands within ~1 % of -eps
        # (z/(z+eps) has a pole at z=-eps; eps=1e-6 on the P.V output is the usual culprit): the
        # reference's own fp32 and fp64 runs then disagree by 1e-2..1e-1 (seed 2 of "mid": 3.9e-2).
        # A 1e-4 parity bar only means something on instances the reference itself resolves, so
        # search the weight seed until the reference's fp32-vs-fp64 gap is < 5e-6 and record it.
        wseed = c["wseed"]
        while True:
            W = ollama.random_weights(cfg, seed=wseed)
            emb = W["embed"][ids]
            ref32 = ref_explicit_llama(cfg, W, emb)
            W64 = ollama.cast_weights(W, torch.float64)
            ref64 = ref_explicit_llama(cfg, W64, emb.double(), target=ref32["idx"])
            gap = nmax(ref32["R_tok"], ref64["R_tok"])
            print(f"  [{name}] wseed={wseed}: reference fp32-vs-fp64 gap {gap:.2e}")
            if gap < 5e-6:
                break
            wseed += 100
        c["wseed"] = wseed
        eff = ref_efficient_llama(cfg, W, ids, "eager")
        eff_sdpa = ref_efficient_llama(cfg, W, ids, "sdpa")
        o32 = ollama.explain(cfg, W, ids=ids, mode="explicit", dtype=torch.float32)
        o64 = ollama.explain(cfg, W, ids=ids, target=ref32["idx"], mode="explicit", dtype=torch.float64)
        oe32 = ollama.explain(cfg, W, ids=ids, mode="efficient", dtype=torch.float32)
        oe64 = ollama.explain(cfg, W, ids=ids, target=eff["idx"], mode="efficient", dtype=torch.float64)
        assert o32["idx"] == ref32["idx"] == eff["idx"], (o32["idx"], ref32["idx"], eff["idx"])
        print(f"  [{name}] idx={ref32['idx']} logit={ref32['logit']:.6f}  sumR={float(ref32['R_tok'].sum()):.6f}")
        print(f"     oracle-explicit fp32 vs ref-explicit fp32 : tok {nmax(o32['R_tok'], ref32['R_tok']):.2e}"
              f"  neuron {nmax(o32['R_emb'], ref32['R_emb']):.2e}")
        print(f"     oracle-explicit fp64 vs ref-explicit fp64 : tok {nmax(o64['R_tok'], ref64['R_tok']):.2e}"
              f"  neuron {nmax(o64['R_emb'], ref64['R_emb']):.2e}")
        print(f"     ref-explicit fp32 vs ref-explicit fp64    : tok {nmax(ref32['R_tok'], ref64['R_tok']):.2e}")
        print(f"     oracle-efficient fp32 vs ref-efficient    : tok {nmax(oe32['R_tok'], eff['R_tok']):.2e}"
              f"  neuron {nmax(oe32['R_emb'], eff['R_emb']):.2e}   (sdpa vs eager {nmax(eff_sdpa['R_tok'], eff['R_tok']):.2e})")
        print(f"     oracle-efficient fp64 vs ref-efficient    : tok {nmax(oe64['R_tok'], eff['R_tok']):.2e}")
        print(f"     ref-efficient vs ref-explicit (fp32)      : tok {nmax(eff['R_tok'], ref32['R_tok']):.2e}")
        print(f"     layer_R oracle vs ref: {max(abs(a - b) for a, b in zip(o32['layer_R'], ref32['layer_R'])):.2e}")
        assert nmax(o32["R_tok"], ref32["R_tok"]) < 2e-5 and nmax(o64["R_tok"], ref64["R_tok"]) < 2e-5
        assert nmax(oe32["R_tok"], eff["R_tok"]) < 2e-5
        fx = dict(cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array([float(v) for v in cfg.values()]),
                  S=S, wseed=c["wseed"], wsum=wsum(W), ids=ids, cond_gap=gap,
                  idx=ref32["idx"], logit=ref32["logit"],
                  exp32_R_tok=ref32["R_tok"], exp64_R_tok=ref64["R_tok"],
                  exp32_layer_R=np.array(ref32["layer_R"]), exp64_layer_R=np.array(ref64["layer_R"]),
                  eff_R_tok=eff["R_tok"], eff_logit=eff["logit"], logits_last=ref32["logits_last"])
        if name != "mid":
            fx["exp64_R_emb"] = ref64["R_emb"].float()
            fx["eff_R_emb"] = eff["R_emb"].float()
        if name == "tiny":   # small enough to carry its weights, so the fixture is self-contained
            fx["W_embed"], fx["W_norm"], fx["W_lm_head"] = W["embed"], W["norm"], W["lm_head"]
            for i, L in enumerate(W["layers"]):
                for k, v in L.items():
                    fx[f"W_l{i}_{k}"] = v
        save(f"llama_{name}.npz", **fx)


if __name__ == "__main__":
    torch.set_num_threads(8)
    golden_rules()
    golden_models()
    print("done")
