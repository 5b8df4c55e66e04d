"""GPU: the drop-in API surface (lxt_amd.explicit / lxt_amd.efficient) against the reference's own
test formulas (ref: tests/test_functional.py, tests/test_rules.py, tests/test_modules.py -- restated,
seeded, on the device) and against the golden fixtures captured from the real reference."""
import math
import warnings
from functools import partial

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.util import nmax, load, t, llama_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lf():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import lxt_amd.explicit.functional as m
    return m


def rn(*s, seed=0, rg=True):
    x = torch.randn(*s, generator=torch.Generator().manual_seed(seed)).cuda()
    return x.requires_grad_() if rg else x


# ------------------------------------------------------------------ ref: tests/test_functional.py
def test_softmax(lf):
    x, R = rn(16, 10, 32, seed=1), rn(16, 10, 32, seed=2, rg=False)
    y_gt = F.softmax(x, -1)
    gt = x.float() * (R - y_gt * R.sum(-1, keepdim=True))
    for inplace in (False, True):
        y = lf.softmax(x, -1, torch.float32, 1.0, inplace)
        assert torch.allclose(y, y_gt, atol=1e-6)
        rel, = torch.autograd.grad(y, x, R)
        assert torch.allclose(gt, rel, rtol=0, atol=1e-5)
    # non-last dim + temperature
    y = lf.softmax(x, 1, None, 2.0)
    assert torch.allclose(y, F.softmax(x / 2.0, 1), atol=1e-6)


def test_matmul(lf):
    eps = 1e-9
    a, b, R = rn(2, 10, 32, seed=1), rn(2, 32, 5, seed=2), rn(2, 10, 5, seed=3, rg=False)
    y_gt = torch.matmul(a, b)
    ga = torch.einsum("bji, bip, bjp -> bji", a, b, R / (2 * y_gt + eps))
    gb = torch.einsum("bji, bip, bjp -> bip", a, b, R / (2 * y_gt + eps))
    y = lf.matmul(a, b, False, eps)
    assert torch.allclose(y, y_gt, atol=1e-5)
    ra, rb = torch.autograd.grad(y, (a, b), R)
    # same z on both sides is what makes this comparison well conditioned: use the kernel's own z
    ga2 = torch.einsum("bji, bip, bjp -> bji", a, b, R / (2 * y.detach() + eps))
    gb2 = torch.einsum("bji, bip, bjp -> bip", a, b, R / (2 * y.detach() + eps))
    assert nmax(ra, ga2) < 1e-5 and nmax(rb, gb2) < 1e-5
    assert nmax(ra, ga) < 1e-2 and nmax(rb, gb) < 1e-2          # random R: ill-conditioned wrt z (SURVEY finding 5)


def test_linear(lf):
    eps = 1e-9
    x, bias, W, R = rn(16, 10, seed=1), rn(5, seed=2, rg=False), rn(5, 10, seed=3), rn(16, 5, seed=4, rg=False)
    y = lf.linear_epsilon(x, W, bias, eps)
    assert torch.allclose(y, F.linear(x, W, bias), atol=1e-5)
    rel, = torch.autograd.grad(y, x, R)
    gt = torch.einsum("ji, bi, bj -> bi", W, x, R / (y.detach() + eps))
    assert nmax(rel, gt) < 1e-5


def test_sum_mean_normalize(lf):
    eps = 1e-9
    a, b, R = rn(16, 10, 32, seed=1), rn(16, 10, 32, seed=2), rn(16, 10, 32, seed=3, rg=False)
    y = lf.add2(a, b, False, eps)
    ra, rb = torch.autograd.grad(y, (a, b), R)
    assert nmax(ra, a * (R / (a + b + eps))) < 1e-5 and nmax(rb, b * (R / (a + b + eps))) < 1e-5
    m = rn(1, 8, 32, seed=4)
    Rm = rn(1, 8, seed=5, rg=False)
    gt = m * (Rm.unsqueeze(-1) / (m.sum(-1).unsqueeze(-1) + eps))
    r1, = torch.autograd.grad(lf.mean(m, -1, True, eps), m, Rm.unsqueeze(-1))
    r2, = torch.autograd.grad(lf.mean(m, -1, False, eps), m, Rm)
    assert nmax(r1, gt) < 1e-5 and nmax(r2, gt) < 1e-5
    x, r = rn(1, 4, 32, seed=6), rn(1, 4, 32, seed=7, rg=False)
    lf.rms_norm_identity(x, rn(32, seed=8, rg=False), 1e-9).backward(r)
    assert torch.allclose(x.grad, r)
    x.grad = None
    lf.normalize(x, p=2, dim=1).backward(r)
    assert torch.allclose(x.grad, r)
    # mul2: uniform split over the operands that require grad
    Ra, Rb = torch.autograd.grad(lf.mul2(a, b), (a, b), R)
    assert torch.allclose(Ra, R / 2) and torch.allclose(Rb, R / 2)
    Ra, = torch.autograd.grad(lf.mul2(a, b.detach()), (a,), R)
    assert torch.allclose(Ra, R)


def test_layernorm_golden(lf):
    fx = load("rules.npz")
    x, w, b, g = (t(fx[k]).cuda() for k in ("ln_x", "ln_w", "ln_b", "ln_g"))
    x.requires_grad_()
    y = lf.layer_norm(x, w, b, 1e-12)
    assert nmax(y, fx["ln_y"]) < 1e-5
    rel, = torch.autograd.grad(y, x, y.detach() * g)
    assert nmax(rel, fx["ln_Rin"]) < 5e-5


# ---------------------------------------------------------------------- ref: tests/test_rules.py
def test_epsilon_rule_equals_functional(lf):
    import lxt_amd.explicit.rules as rules
    x, W, bias, R = rn(1, 8, seed=1), rn(8, 8, seed=2, rg=False), rn(8, seed=3, rg=False), rn(1, 8, seed=4, rg=False)
    y = lf.linear_epsilon(x, W, bias)
    gt, = torch.autograd.grad(y, x, R)
    y2 = rules.EpsilonRule(partial(F.linear, weight=W, bias=bias))(x)       # generic VJP path
    r2, = torch.autograd.grad(y2, x, R)
    assert torch.allclose(gt, r2, rtol=0, atol=1e-3)
    lin = nn.Linear(8, 8).cuda()
    y3 = rules.EpsilonRule(lin, 1e-6)(x)                                    # fused nn.Linear path
    y4 = lf.linear_epsilon(x, lin.weight, lin.bias, 1e-6)
    assert torch.equal(y3, y4)
    # uniform / identity / stop rules
    a, b = rn(4, 6, seed=5), rn(4, 6, seed=6)

    class Mul(nn.Module):
        def forward(self, p, q):
            return p * q
    Ra, Rb = torch.autograd.grad(rules.UniformRule(Mul())(a, b), (a, b), R[:, :6].expand(4, 6).contiguous())
    assert torch.allclose(Ra, Rb)
    z = rules.IdentityRule(nn.SiLU())(a)
    ri, = torch.autograd.grad(z, a, torch.ones_like(z))
    assert torch.equal(ri, torch.ones_like(z))


def test_uniform_epsilon_golden(lf):
    import lxt_amd.explicit.rules as rules
    fx = load("rules.npz")
    p, v, g = (t(fx[k]).cuda().requires_grad_() for k in ("pv_p", "pv_v", "pv_g"))

    class AV(nn.Module):
        def forward(self, a_, v_):
            return torch.matmul(a_, v_)
    o = rules.UniformEpsilonRule(AV())(p, v)
    Rp, Rv = torch.autograd.grad(o, (p, v), o.detach() * g.detach())
    assert nmax(Rp, fx["pv_Rp"]) < 2e-5 and nmax(Rv, fx["pv_Rv"]) < 2e-5


# -------------------------------------------------------------------- ref: tests/test_modules.py
def test_modules_and_composite(lf):
    import lxt_amd.explicit.modules as lm
    from lxt_amd.explicit import Composite
    x = rn(4, 50, 96, seed=1, rg=False)
    ln = nn.LayerNorm(96).cuda()
    nn.init.normal_(ln.weight); nn.init.normal_(ln.bias)
    new = lm.INIT_MODULE_MAPPING[lm.LayerNormEpsilon](ln, lm.LayerNormEpsilon)
    assert torch.allclose(new(x), ln(x), atol=1e-5)
    lin = nn.Linear(96, 40).cuda()
    newl = lm.INIT_MODULE_MAPPING[lm.LinearEpsilon](lin, lm.LinearEpsilon)
    assert torch.allclose(newl(x), lin(x), atol=1e-5) and newl.weight is lin.weight
    model = nn.Sequential(nn.Linear(96, 64), nn.LayerNorm(64), nn.Linear(64, 8)).cuda()
    ref_out = model(x)
    comp = Composite({nn.Linear: lm.LinearEpsilon, nn.LayerNorm: lm.LayerNormEpsilon})
    comp.register(model)
    assert isinstance(model[0], lm.LinearEpsilon) and all(not p.requires_grad for p in model.parameters())
    xin = x.clone().requires_grad_()
    out = model(xin)
    assert torch.allclose(out, ref_out, atol=1e-4)
    out[0, 0, 3].backward(out[0, 0, 3].detach())          # explicit protocol: seed with the logit
    assert torch.isfinite(xin.grad).all() and xin.grad[1:].abs().max() == 0
    comp.remove()
    assert type(model[0]) is nn.Linear


def test_multihead_attention_cp_golden():
    """lxt.explicit.modules.MultiheadAttention_CP (CP-LRP attention for ViTs) against outputs, attention
    weights and input relevance captured from the real reference (tests/golden/make_golden_mha.py): additive float
    attn_mask, boolean key_padding_mask, no mask; EpsilonRule on v_proj / out_proj as in the reference's own test"""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import lxt_amd.explicit.modules as lm
    import lxt_amd.explicit.rules as rules
    fx = load("mha_cp.npz")
    kws = dict(mask=dict(attn_mask=t(fx["attn_mask"]).cuda()), kpm=dict(key_padding_mask=t(fx["key_padding_mask"]).cuda()), none={})
    worst = 0.0
    for name, kw in kws.items():
        torch.manual_seed(21)
        gt = nn.MultiheadAttention(256, 4, batch_first=True).eval()
        assert abs(float(sum(p.double().abs().sum() for p in gt.parameters())) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
        gt = gt.cuda()
        layer = lm.INIT_MODULE_MAPPING[lm.MultiheadAttention_CP](gt, lm.MultiheadAttention_CP)
        layer.v_proj = rules.EpsilonRule(layer.v_proj)
        layer.out_proj = rules.EpsilonRule(layer.out_proj)
        x = t(fx[f"{name}_x"]).cuda().requires_grad_()
        y, attn = layer(x, x, x, **kw)
        assert nmax(y, fx[f"{name}_y"]) < 1e-5 and nmax(attn, fx[f"{name}_attn"]) < 1e-5
        y.backward(y)
        err = max(nmax(x.grad, fx[f"{name}_R"]), nmax(x.grad, fx[f"{name}_R_fp64"]))
        print(f"[MultiheadAttention_CP/{name}] y {nmax(y, fx[f'{name}_y']):.2e} attn {nmax(attn, fx[f'{name}_attn']):.2e} R_in {err:.2e}")
        worst = max(worst, err)
        assert abs(float(x.grad.sum()) - float(t(fx[f"{name}_R"]).sum())) < 1e-3 * abs(float(t(fx[f"{name}_R"]).sum()))
    assert worst < 1e-4


# --------------------------------------------------------------------------- efficient drop-in path
def _hf_llama(cfg, W, attn_impl, rope_parameters=None):
    from transformers import LlamaConfig, LlamaForCausalLM
    hc = LlamaConfig(hidden_size=cfg["hidden"], intermediate_size=cfg["inter"], num_hidden_layers=cfg["n_layers"],
                     num_attention_heads=cfg["n_heads"], num_key_value_heads=cfg["n_kv"], head_dim=cfg["head_dim"],
                     vocab_size=cfg["vocab"], rms_norm_eps=cfg["rms_eps"], max_position_embeddings=4096,
                     rope_parameters=rope_parameters or dict(rope_type="default", rope_theta=cfg["rope_theta"]),
                     tie_word_embeddings=False, attn_implementation=attn_impl)
    model = LlamaForCausalLM(hc).eval()
    with torch.no_grad():
        model.model.embed_tokens.weight.copy_(W["embed"]); model.model.norm.weight.copy_(W["norm"])
        model.lm_head.weight.copy_(W["lm_head"])
        for L, Lw in zip(model.model.layers, W["layers"]):
            L.input_layernorm.weight.copy_(Lw["ln1"]); L.post_attention_layernorm.weight.copy_(Lw["ln2"])
            L.self_attn.q_proj.weight.copy_(Lw["wq"]); L.self_attn.k_proj.weight.copy_(Lw["wk"])
            L.self_attn.v_proj.weight.copy_(Lw["wv"]); L.self_attn.o_proj.weight.copy_(Lw["wo"])
            L.mlp.gate_proj.weight.copy_(Lw["wg"]); L.mlp.up_proj.weight.copy_(Lw["wu"]); L.mlp.down_proj.weight.copy_(Lw["wd"])
    for p in model.parameters():
        p.requires_grad_(False)
    return model.cuda()


@pytest.mark.parametrize("name", ["mid", "d128"])
def test_monkey_patch_llama_user_protocol(name):
    """the reference's quickstart protocol, unchanged, on a HF model patched by lxt_amd:
    model(inputs_embeds=e.requires_grad_()).logits[0,-1,i].backward(); R = (e*e.grad).sum(-1)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from transformers.models.llama import modeling_llama
    from lxt_amd.efficient import monkey_patch
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkey_patch(modeling_llama)
        monkey_patch(modeling_llama)                           # idempotent: warns, does not raise
    cfg, W, ids, fx = llama_case(name)
    for impl in ("eager", "sdpa"):                             # both dispatch to the HIP attention now
        model = _hf_llama(cfg, W, impl)
        e = model.get_input_embeddings()(ids[None].cuda()).requires_grad_()
        logits = model(inputs_embeds=e, use_cache=False).logits
        idx = int(logits[0, -1].argmax())
        assert idx == int(fx["idx"])
        logits[0, -1, idx].backward()
        R = (e * e.grad).float().sum(-1)[0]
        err = nmax(R, fx["eff_R_tok"])
        print(f"[monkey_patch {name}/{impl}] tok vs reference lxt.efficient {err:.2e}")
        assert err < 1e-4
        # engine == drop-in path (same kernels, different host sequencing)
    import lxt_amd.engine as E
    eng = E.LlamaLRP.from_hf(model, mode="efficient", max_seq=512)
    out = eng.explain(ids[None])
    assert nmax(out["R_tok"][0], R) < 1e-5


def test_unsupported_module_raises():
    import types
    from lxt_amd.efficient import monkey_patch
    with pytest.raises(ValueError, match="not yet supported"):
        monkey_patch(types.ModuleType("some.random.module"))


def _patch_llama():
    from transformers.models.llama import modeling_llama
    from lxt_amd.efficient import monkey_patch
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkey_patch(modeling_llama)


def test_patched_linear_leaves_foreign_cuda_modules_alone():
    """VERDICT r1 weak #6 / ADVICE: after monkey_patch an unrelated CUDA nn.Linear (never part of an explained model) must be
    bit-identical to ATen and keep its parameter gradients; an owned Linear whose Parameter is REPLACED (resize / tie /
    adapter loading keep _version 0) must not reuse a stale W^T copy"""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    _patch_llama()
    lin = nn.Linear(96, 80).cuda()
    x = rn(7, 96, seed=3)
    y = lin(x)
    assert torch.equal(y, F.linear(x, lin.weight, lin.bias))
    y.sum().backward()
    assert lin.weight.grad is not None and x.grad is not None
    from lxt_amd.efficient import adopt
    own = adopt(nn.Linear(96, 80, bias=False).cuda().requires_grad_(False))
    x2 = rn(7, 96, seed=4)
    y1 = own(x2)
    assert nmax(y1, F.linear(x2.detach(), own.weight)) < 1e-5
    y1.sum().backward()
    g1 = x2.grad.clone()
    assert nmax(g1, own.weight.sum(0)[None].expand(7, 96)) < 1e-5
    own.weight = nn.Parameter(torch.randn(64, 96, device="cuda"), requires_grad=False)       # new storage AND new shape
    x3 = rn(7, 96, seed=5)
    y2 = own(x3)
    assert y2.shape == (7, 64)
    y2.sum().backward()
    assert nmax(x3.grad, own.weight.sum(0)[None].expand(7, 96)) < 1e-5


def test_engine_from_hf_llama3_rope_scaling():
    """Llama-3.1 / 3.2 checkpoints use rope_type='llama3'; LlamaLRP.from_hf must reproduce HF's logits and
    the relevance of the drop-in path with those frequencies (it silently used plain RoPE before)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    _patch_llama()
    cfg, W, ids, fx = llama_case("mid")
    rp = dict(rope_type="llama3", rope_theta=cfg["rope_theta"], factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
              original_max_position_embeddings=64)        # small original length: every band of the blend is exercised at S=128
    model = _hf_llama(cfg, W, "eager", rope_parameters=rp)
    e = model.get_input_embeddings()(ids[None].cuda()).requires_grad_()
    logits = model(inputs_embeds=e, use_cache=False).logits[0, -1]
    idx = int(logits.argmax())
    logits[idx].backward()
    R = (e * e.grad).float().sum(-1)[0]
    import lxt_amd.engine as E
    eng = E.LlamaLRP.from_hf(model, mode="efficient", max_seq=512)
    assert "inv_freq" in eng.cfg
    out = eng.explain(ids[None])
    assert int(out["idx"][0]) == idx and nmax(out["logits"][0], logits) < 1e-5
    assert nmax(out["R_tok"][0], R) < 1e-5
    plain = E.LlamaLRP(dict(cfg), W, dtype=torch.float32, mode="efficient", max_seq=512).explain(ids[None])
    assert nmax(plain["logits"][0], logits) > 1e-3            # the scaling matters on this instance
    # ... and against the oracle's own restatement of the llama3 frequencies, fp64
    from oracle import llama as ol
    ocfg = dict(cfg, rope_scaling={k: v for k, v in rp.items() if k != "rope_theta"})
    ref = ol.explain(ocfg, W, ids=ids, target=idx, mode="efficient", dtype=torch.float64)
    assert nmax(out["R_tok"][0], ref["R_tok"]) < 1e-4


def test_checkpointing_and_retain_grad_protocols():
    """the two memory / latent-relevance protocols of the reference's docs on the drop-in path:
    docs/source/quickstart.rst:84-90 (params frozen, model.train() + gradient_checkpointing_enable(): 2x forward, the patched
    Dropout keeps p = 0) and docs/source/latent-feature-attribution-efficient.rst:50-56,86-90 (forward hooks + retain_grad on
    every decoder layer's output, relevance trace = (output * output.grad).sum(-1)) -- against the plain run, the fused
    engine's per-layer trace and the oracle"""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    _patch_llama()
    cfg, W, ids, fx = llama_case("mid")
    model = _hf_llama(cfg, W, "sdpa")

    def hook(module, inp, output):
        output = output[0] if isinstance(output, tuple) else output
        module.output = output
        if output.requires_grad:
            output.retain_grad()
    for layer in model.model.layers:
        layer.register_forward_hook(hook)

    def run():
        e = model.get_input_embeddings()(ids[None].cuda()).requires_grad_()
        logits = model(inputs_embeds=e, use_cache=False).logits
        mx, mi = torch.max(logits[:, -1, :], dim=-1)
        mx.backward(mx)                                              # the doc's seeding: relevance = the logit itself
        R = (e * e.grad).float().sum(-1)[0]
        trace = torch.stack([(l.output * l.output.grad).float().sum(-1)[0] for l in model.model.layers])
        return int(mi), float(mx), R, trace
    idx0, z0, R0, T0 = run()
    model.train()
    model.gradient_checkpointing_enable()
    idx1, z1, R1, T1 = run()
    assert idx0 == idx1 == int(fx["idx"])
    assert nmax(R1, R0) < 1e-6 and nmax(T1, T0) < 1e-6            # checkpointed re-forward reproduces the plain run
    assert nmax(R0 / z0, fx["eff_R_tok"]) < 1e-4                     # seeded with the logit value -> relevance scales by it
    import lxt_amd.engine as E
    out = E.LlamaLRP.from_hf(model, mode="efficient", max_seq=512).explain(ids[None], layer_relevance=True)
    assert nmax(T0.sum(-1) / z0, out["layer_R"][1:, 0]) < 1e-4       # per-layer latent relevance == the fused engine's trace
    from oracle import llama as ol
    ref = ol.explain(cfg, W, ids=ids, target=idx0, mode="efficient", dtype=torch.float64)
    assert nmax(T0.sum(-1) / z0, torch.tensor(ref["layer_R"][1:])) < 1e-4


def test_explicit_rules_under_checkpoint():
    """ref: lxt/explicit/rules.py:192-195 -- inside torch.utils.checkpoint's no-grad first pass the rule just evaluates the
    function; the recomputation pass applies the rule.  The wrapped-module API must give the same relevance either way."""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from torch.utils.checkpoint import checkpoint
    import lxt_amd.explicit.rules as rules
    lin = nn.Linear(64, 48).cuda().requires_grad_(False)
    mod = rules.EpsilonRule(lin, epsilon=1e-6)
    x = rn(5, 64, seed=11)
    g = rn(5, 48, seed=12, rg=False)
    y = mod(x)
    y.backward(y.detach() * g)
    R_plain = x.grad.clone()
    x.grad = None
    y2 = checkpoint(mod, x, use_reentrant=True)
    y2.backward(y2.detach() * g)
    assert nmax(x.grad, R_plain) < 1e-6


def test_conservation_check_mode_covers_wrapped_rules():
    """with check.conservation_check() every rule's backward hands sum(R_out) spread uniformly over its inputs
    (ref: lxt/explicit/functional.py:10-37 decorates the rules of lxt/explicit/rules.py:75,119,211,271,413 too) -- through
    the wrapped-module API a Composite registers"""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import lxt_amd.explicit.rules as rules
    from lxt_amd.explicit.check import conservation_check
    cases = [(rules.UniformEpsilonRule(torch.matmul), (rn(6, 8, seed=1), rn(8, 5, seed=2))),
             (rules.EpsilonRule(torch.matmul), (rn(6, 8, seed=3), rn(8, 5, seed=4))),
             (rules.UniformRule(torch.mul), (rn(6, 8, seed=5), rn(6, 8, seed=6))),
             (rules.IdentityRule(nn.SiLU()), (rn(6, 8, seed=7),)),
             (rules.EpsilonRule(nn.Linear(8, 5).cuda().requires_grad_(False)), (rn(6, 8, seed=8),))]
    for mod, inputs in cases:
        with conservation_check():
            y = mod(*inputs)
            y.backward(torch.ones_like(y))
        total = sum(float(x.grad.sum()) for x in inputs)
        assert abs(total - y.numel()) < 1e-3 * y.numel(), (type(mod).__name__, total, y.numel())
        for x in inputs:
            assert float(x.grad.max() - x.grad.min()) == 0.0          # uniform spread, not the rule's own relevance
            x.grad = None
        if isinstance(mod, (rules.EpsilonRule, rules.UniformEpsilonRule)):
            y = mod(*inputs)                                          # flag off again: the rule's real (non-uniform) relevance
            y.backward(torch.ones_like(y))
            assert any(float(x.grad.max() - x.grad.min()) != 0.0 for x in inputs)


def test_import_order_package_before_torch():
    """`import lxt_amd` AHEAD of `import torch` in a fresh interpreter: the binding loads torch's HIP runtime before liblrp_hip.so, so both
    use ONE libamdhip64 (PyTorch wheels ship their own; bound to the system copy, every launch failed with hipErrorNoDevice)"""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import lxt_amd.ops as ops\n"
            "import torch\n"
            "x = torch.randn(64, 96, device='cuda').bfloat16()\n"
            "assert torch.equal(ops.transpose(x), x.t().contiguous())\n"
            "print('ok')\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
