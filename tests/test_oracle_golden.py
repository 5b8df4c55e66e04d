"""CPU: the oracle (oracle/) against the golden vectors captured from the real reference
(tests/golden/make_golden.py) and against the reference tests' closed forms
(/root/reference tests/test_functional.py:16,41-42,70,91-92,117; tests/test_rules.py:9-24)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import rules as R
from oracle import llama as ol
from tests.util import load, t, nmax, llama_case


@pytest.fixture(scope="module")
def fx():
    return load("rules.npz")


@pytest.mark.parametrize("tag", ["c1", "toy", "mid"])
@pytest.mark.parametrize("eps_tag,eps", [("f", 1e-6), ("r", 1e-8)])
def test_linear_epsilon_bitexact(fx, tag, eps_tag, eps):
    # BASELINE config 1: Linear(768->768) eps-rule, batch 1, CPU, bit-compare
    x, W, b, g = (t(fx[f"lin_{tag}_{k}"]) for k in "xWbg")
    z = F.linear(x, W, b)
    _, Rin = R.linear_epsilon(x, W, b, z * g, eps)
    assert torch.equal(Rin, t(fx[f"lin_{tag}_{eps_tag}_Rin"]))


def test_linear_closed_form():
    # reference tests/test_functional.py:57-76 (einsum of Eq. 8)
    g = torch.Generator().manual_seed(3)
    x, bias, W = torch.randn(16, 10, generator=g), torch.randn(5, generator=g), torch.randn(5, 10, generator=g)
    Rout = torch.randn(16, 5, generator=g)
    y = F.linear(x, W, bias)
    gt = torch.einsum("ji, bi, bj -> bi", W, x, Rout / (y + 1e-9))
    assert torch.allclose(R.linear_epsilon(x, W, bias, Rout, 1e-9)[1], gt, rtol=0, atol=1e-3)


def test_matmul(fx):
    a, b, g = t(fx["mm_a"]), t(fx["mm_b"]), t(fx["mm_g"])
    o, Ra, Rb = R.matmul(a, b, torch.matmul(a, b) * g, 1e-8)
    assert nmax(Ra, fx["mm_Ra"]) < 1e-6 and nmax(Rb, fx["mm_Rb"]) < 1e-6
    # closed form, reference tests/test_functional.py:41-42
    Rout = torch.randn(2, 3, 10, 7, generator=torch.Generator().manual_seed(1))
    gt_a = torch.einsum("hbji, hbip, hbjp -> hbji", a, b, Rout / (2 * o + 1e-9))
    assert torch.allclose(R.matmul(a, b, Rout, 1e-9)[1], gt_a, rtol=0, atol=1e-4)


def test_softmax(fx):
    x, g = t(fx["sm_x"]), t(fx["sm_g"])
    p = F.softmax(x, -1)
    p2, Rx = R.softmax(x, p * g)
    assert nmax(p2, fx["sm_p"]) < 1e-7 and nmax(Rx, fx["sm_Rx"]) < 1e-6
    assert torch.isfinite(Rx).all()


def test_add2_mul2_mean(fx):
    a, b, g = t(fx["add_a"]), t(fx["add_b"]), t(fx["add_g"])
    _, Ra, Rb = R.add2(a, b, (a + b) * g, 1e-8)
    assert nmax(Ra, fx["add_Ra"]) < 1e-6 and nmax(Rb, fx["add_Rb"]) < 1e-6
    _, Ra, Rb = R.mul2(t(fx["mul_a"]), t(fx["mul_b"]), t(fx["mul_R"]))
    assert torch.equal(Ra, t(fx["mul_Ra"])) and torch.equal(Rb, t(fx["mul_Rb"]))
    assert torch.equal(R.mul2(t(fx["mul_a"]), t(fx["mul_b"]), t(fx["mul_R"]), True, False)[1], t(fx["mul_Ra_const"]))
    assert nmax(R.mean(t(fx["mean_a"]), t(fx["mean_R"]), -1, True, 1e-6)[1], fx["mean_Rin"]) < 1e-6


def test_norms(fx):
    y, Rin = R.rms_norm_identity(t(fx["rms_x"]), t(fx["rms_w"]), 1e-5, t(fx["rms_R"]))
    assert torch.equal(y, t(fx["rms_y"])) and torch.equal(Rin, t(fx["rms_Rin"]))   # pure pass-through
    x, w, b, g = t(fx["ln_x"]), t(fx["ln_w"]), t(fx["ln_b"]), t(fx["ln_g"])
    y = F.layer_norm(x, (48,), w, b, 1e-12)
    y2, Rin = R.layer_norm(x, w, b, 1e-12, y * g, 1e-6)
    assert nmax(y2, fx["ln_y"]) < 1e-6 and nmax(Rin, fx["ln_Rin"]) < 1e-5


def test_uniform_epsilon_and_efficient_primitives(fx):
    p, v, g = t(fx["pv_p"]), t(fx["pv_v"]), t(fx["pv_g"])
    _, Rp, Rv = R.uniform_epsilon_matmul(p, v, torch.matmul(p, v) * g, 1e-6)
    assert nmax(Rp, fx["pv_Rp"]) < 1e-6 and nmax(Rv, fx["pv_Rv"]) < 1e-6
    x, G = t(fx["act_x"]), t(fx["act_G"])
    y, Gi = R.identity_rule_implicit(F.silu, x, G)
    assert torch.equal(y, t(fx["act_silu_y"])) and nmax(Gi, fx["act_silu_Gin"]) < 1e-7
    y, Gi = R.identity_rule_implicit(lambda z: F.gelu(z, approximate="tanh"), x, G)
    assert nmax(Gi, fx["act_gelut_Gin"]) < 1e-7
    assert torch.equal(R.divide_gradient(x, G, 4)[1], G / 4)


@pytest.mark.parametrize("name", ["tiny", "mid", "d128"])
def test_llama_oracle_vs_reference(name):
    cfg, W, ids, fx = llama_case(name)
    o32 = ol.explain(cfg, W, ids=ids, mode="explicit", dtype=torch.float32)
    assert o32["idx"] == int(fx["idx"]) and abs(o32["logit"] - float(fx["logit"])) < 1e-5
    assert nmax(o32["R_tok"], fx["exp32_R_tok"]) < 5e-6          # vs reference explicit fp32
    assert nmax(o32["layer_R"], fx["exp32_layer_R"]) < 5e-6      # latent per-layer relevance sums
    o64 = ol.explain(cfg, W, ids=ids, target=int(fx["idx"]), mode="explicit", dtype=torch.float64)
    assert nmax(o64["R_tok"], fx["exp64_R_tok"]) < 5e-6          # vs reference explicit fp64
    if "exp64_R_emb" in fx:
        assert nmax(o64["R_emb"], fx["exp64_R_emb"]) < 5e-6      # per-neuron
    oe = ol.explain(cfg, W, ids=ids, mode="efficient", dtype=torch.float32)
    assert nmax(oe["R_tok"], fx["eff_R_tok"]) < 5e-6             # vs reference lxt.efficient
    # conservation sanity: sum of token relevance stays O(logit)
    assert abs(float(o64["R_tok"].sum())) < 10 * abs(float(fx["logit"])) + 1.0


def test_oracle_kv_chunk_equals_unchunked():
    """the head-chunked attention of the oracle (kv_chunk: scores / probabilities recomputed per kv group in the backward instead of kept
    for every layer -- what lets the fp64 oracle run BASELINE config 5's 32 / 8 heads at S = 4096 on a 62-GB host) is the same arithmetic
    as the un-chunked oracle that the fixtures pin: bit-identical relevance in both modes"""
    import torch
    from oracle import llama as ol
    cfg = dict(hidden=64, inter=128, n_layers=2, n_heads=8, n_kv=4, head_dim=16, vocab=100, rope_theta=10000.0, rms_eps=1e-5)
    W = ol.random_weights(cfg, seed=3)
    ids = torch.randint(0, 100, (40,), generator=torch.Generator().manual_seed(1))
    for mode in ("explicit", "efficient"):
        a = ol.explain(cfg, W, ids=ids, mode=mode, dtype=torch.float64)
        for kc in (1, 3):
            b = ol.explain(cfg, W, ids=ids, mode=mode, dtype=torch.float64, kv_chunk=kc)
            assert a["idx"] == b["idx"] and torch.equal(a["R_tok"], b["R_tok"]) and a["layer_R"] == b["layer_R"]
