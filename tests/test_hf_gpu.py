"""GPU: full-model relevance of HuggingFace models patched by lxt_amd.efficient.monkey_patch against
(1) the fixtures captured from the real reference's primitives (tests/golden/make_golden_hf.py) and
(2) the instance-level CPU oracle (oracle/hf_efficient.py) run live in fp64.
BASELINE config 2: BERT-base, S=128, fp32.  Gemma3 text tower: sliding + global layers, q/k-norm,
(1+w) RMSNorm, gelu-tanh.  Tolerance 1e-4 normalised max error per token (fp32)."""
import warnings

import pytest
import torch

from oracle import hf_efficient as oh
from tests.golden.hf_models import build_bert, build_gemma3, wsum
from tests.util import nmax, load, t

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")


def test_bert_base_full_model_relevance():
    _need_gpu()
    fx = load("bert_base.npz")
    ids = t(fx["ids"])
    ref64 = oh.explain_classifier(oh.patch_instance(build_bert(seed=0, attn="eager").double()), ids, target=int(fx["idx"]))
    from transformers.models.bert import modeling_bert
    from lxt_amd.efficient import monkey_patch
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkey_patch(modeling_bert)
    for impl in ("eager", "sdpa"):
        model = build_bert(seed=0, attn=impl)
        assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
        for p in model.parameters():
            p.requires_grad_(False)
        model = model.cuda()
        e = model.get_input_embeddings()(ids[None].cuda()).requires_grad_()
        logits = model(inputs_embeds=e).logits[0]
        idx = int(logits.argmax())
        assert idx == int(fx["idx"]) and abs(float(logits[idx]) - float(fx["logit"])) < 1e-4
        logits[idx].backward()
        R = (e * e.grad)[0].sum(-1)
        e1, e2 = nmax(R, fx["R_tok"]), nmax(R, ref64["R_tok"])
        print(f"[bert-base/{impl}] tok vs reference {e1:.2e} | vs oracle fp64 {e2:.2e}")
        assert e1 < 1e-4 and e2 < 1e-4


def test_gemma3_text_full_model_relevance():
    _need_gpu()
    fx = load("gemma3_tiny.npz")
    ids = t(fx["ids"])
    ref64 = oh.explain_causal_lm(oh.patch_instance(build_gemma3(seed=3, attn="eager").double()), ids, target=int(fx["idx"]))
    from transformers.models.gemma3 import modeling_gemma3
    from lxt_amd.efficient import monkey_patch
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkey_patch(modeling_gemma3)
    for impl in ("eager", "sdpa"):
        model = build_gemma3(seed=3, attn=impl)
        assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
        for p in model.parameters():
            p.requires_grad_(False)
        model = model.cuda()
        e = model.get_input_embeddings()(ids[None].cuda()).requires_grad_()
        last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
        idx = int(last.argmax())
        assert idx == int(fx["idx"])
        last[idx].backward()
        R = (e * e.grad)[0].sum(-1)
        e1, e2 = nmax(R, fx["R_tok"]), nmax(R, ref64["R_tok"])
        print(f"[gemma3/{impl}] tok vs reference {e1:.2e} | vs oracle fp64 {e2:.2e}")
        assert e1 < 1e-4 and e2 < 1e-4


@pytest.mark.parametrize("which", ["llama_cp", "qwen2", "qwen3", "gpt2", "qwen2_padded", "gemma3_mm", "gemma3_mm_4bdims", "mini_vit"])
def test_model_family_maps(which):
    """CP-LRP map (llama) and the qwen2 / qwen3 / gpt2 AttnLRP maps against fixtures captured from the
    reference's own maps; one fresh process per family (class-level patches are process-global)."""
    _need_gpu()
    import os, subprocess, sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "hf_family_worker.py"), which], capture_output=True, text=True,
                       timeout=600, cwd=root)
    print(r.stdout[-600:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_bert_base_explicit_full_model_relevance():
    """BASELINE config 2 in lxt.explicit semantics (ref lxt/explicit/models/bert.py:60-65,249-253,338-373,396): nn.Linear eps rule
    with bias, LayerNormEpsilon, lf.matmul (R/(2 O + eps)) on BOTH attention contractions, add2 residuals, GELU / Tanh identity
    rules -- against the relevance the reference's own Functions produced (bert_base_explicit.npz, fp32 and fp64) and the fp64
    oracle.  Two product paths: (1) the SAME composition code the fixture was made with, over lxt_amd.explicit.{functional,rules}
    (the drop-in claim of the explicit API), (2) an unmodified HF BertForSequenceClassification re-wired in place by
    lxt_amd.explicit.models.bert.attnlrp.  Bar: 1e-4, or 3x the reference's own fp32-vs-fp64 gap on this instance (the explicit
    stabilisers have poles)."""
    _need_gpu()
    import lxt_amd.explicit.functional as lf
    import lxt_amd.explicit.rules as rules
    from lxt_amd.explicit.models import bert as xb
    from oracle import bert as ob
    from tests.golden import bert_explicit_compose as C
    fx = load("bert_base_explicit.npz")
    ids = t(fx["ids"])
    model = build_bert(seed=0, attn="eager")
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    bar = max(1e-4, 3 * float(fx["cond_gap"]))
    o64 = ob.explain(C.weights_from_hf(model, torch.float64), ids, target=int(fx["idx"]), dtype=torch.float64)
    assert nmax(o64["R_tok"], fx["R_tok_fp64"]) < 1e-9
    # (1) same composition, HIP backend
    W = C.weights_from_hf(model, torch.float32, device="cuda")
    r = C.explain(lf, rules, W, ids[None].cuda())
    assert r["idx"] == int(fx["idx"]) and abs(r["logit"] - float(fx["logit"])) < 1e-4
    e1 = nmax(r["R_tok"], fx["R_tok_fp64"])
    print(f"[bert-base explicit / composed over lxt_amd.explicit] token vs reference fp64 {e1:.2e} | vs reference fp32 "
          f"{nmax(r['R_tok'], fx['R_tok']):.2e} | neuron {nmax(r['R_emb'], fx['R_emb_fp64']):.2e} (reference's own gap {float(fx['cond_gap']):.1e})")
    assert e1 < bar
    # (2) an unmodified HF instance re-wired in place -- in a fresh process: the efficient-mode tests above patch BERT's classes
    # process-wide (as the reference's monkey_patch does), the explicit wiring is meant for an un-patched transformers
    import os, subprocess, sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "hf_family_worker.py"), "bert_explicit"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    print(r.stdout[-600:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_bert_explicit_padded_batch():
    """explicit wiring on a right-padded batch: the attention mask reaches the custom attention function"""
    _need_gpu()
    import os, subprocess, sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "hf_family_worker.py"), "bert_explicit_padded"], capture_output=True,
                       text=True, timeout=600, cwd=root)
    print(r.stdout[-900:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]



def test_llama_8b_dims_dropin_fp32_parity_and_bf16_like_the_reference_in_bf16():
    """Llama-3-8B layer dimensions (H 4096, I 14336, 32 / 8 heads of d = 128), 8 layers, S = 512, random init, against the REAL reference run on
    the CPU in fp32 and in bf16 (llama_bf16_depth.npz, tests/golden/make_golden_llama_bf16_depth.py).
      * fp32 drop-in path (HF model under lxt_amd.efficient.monkey_patch, autograd-driven) and fp32 fused engine vs the reference fp32: < 1e-4;
      * bf16: the drop-in path keeps HF's own bf16 forward and bf16 autograd, exactly like the reference run in bf16 -- on a deep random-init
        model that arithmetic is noisy (the reference's bf16 vs its own fp32: normalised max error 1.0e-1, cosine 0.989 here; at 32 layers /
        S = 2048 the drop-in path's cosine against fp32 is 0.65, tools/llama_dropin_bench.py).  The drop-in path must be no further from
        the reference's fp32 than 3 x the reference's own bf16 run (it IS that arithmetic); the fused bf16 engine (fp32 statistics and
        accumulators between its kernels) is held to ITS OWN bar against the reference's fp32: normalised max error <= 2e-2, cosine >= 0.9995
        (measured 5.4e-3 / 0.99993; VERDICT r4 "what's weak" 4 -- not through the 3x-reference-bf16 rule, which would allow 0.3)."""
    _need_gpu()
    from tests.golden.hf_models import build_llama_8bdims
    from transformers.models.llama import modeling_llama
    from lxt_amd.efficient import monkey_patch
    import lxt_amd.engine as E
    fx = load("llama_bf16_depth.npz")
    ids = t(fx["ids"]).long()
    idx = int(fx["idx"])
    R32 = torch.as_tensor(fx["R_tok_fp32"]).double()
    model = build_llama_8bdims(layers=int(fx["layers"]))
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-9 * float(fx["wsum"]), "seeded weights did not regenerate"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkey_patch(modeling_llama)
    for p_ in model.parameters():
        p_.requires_grad_(False)
    model = model.cuda()

    def dropin(m):
        e = m.get_input_embeddings()(ids[None].cuda()).detach().requires_grad_()
        last = m(inputs_embeds=e, use_cache=False).logits[0, -1]
        last[idx].backward()
        return float(last[idx]), (e * e.grad)[0].float().sum(-1).double().cpu()

    def stats(R):
        return nmax(R, R32), float(torch.nn.functional.cosine_similarity(R, R32, dim=0))
    lg, Rd32 = dropin(model)
    e_d32 = nmax(Rd32, R32)
    eng32 = E.LlamaLRP.from_hf(model, mode="efficient", max_seq=512, dtype=torch.float32)
    e_e32 = nmax(eng32.explain(ids[None], target=torch.tensor([idx]))["R_tok"][0].double().cpu(), R32)
    eng32.release()
    print(f"[llama 8B dims, 8 layers, S=512, fp32] drop-in vs REFERENCE fp32 {e_d32:.2e} (logit {lg:+.5f} vs {float(fx['logit']):+.5f}) | fused engine {e_e32:.2e}")
    assert e_d32 < 1e-4 and e_e32 < 1e-4
    mb = model.to(torch.bfloat16)
    _, Rd16 = dropin(mb)
    eng16 = E.LlamaLRP.from_hf(mb, mode="efficient", max_seq=512)
    Re16 = eng16.explain(ids[None], target=torch.tensor([idx]))["R_tok"][0].double().cpu()
    (n_d, c_d), (n_e, c_e) = stats(Rd16), stats(Re16)
    n_r, c_r = float(fx["ref_bf16_nmax"]), float(fx["ref_bf16_cos"])
    print(f"[llama 8B dims, 8 layers, S=512, bf16 vs the reference's fp32] reference in bf16 (CPU): {n_r:.2e}, cosine {c_r:.5f} | drop-in path: {n_d:.2e}, "
          f"cosine {c_d:.5f} | fused engine: {n_e:.2e}, cosine {c_e:.5f}")
    assert n_d <= 3 * n_r and (1 - c_d) <= 3 * (1 - c_r)
    assert n_e <= 2e-2 and c_e >= 0.9995, (n_e, c_e)


@pytest.mark.parametrize("name", ["tiny", "mid", "d128"])
@pytest.mark.parametrize("impl", ["eager", "sdpa"])
def test_llama_explicit_composite_dropin(name, impl):
    """SURVEY 8 row a16, explicit Llama as a DROP-IN: `lxt_amd.explicit.models.llama.attnlrp.register(hf_model)` (ref
    lxt/explicit/models/llama.py:83-93 and the splice sites :226-260,:273-281,:379-391,:481-488) on an unmodified HF LlamaForCausalLM, the explicit
    protocol of examples/paper/llama.py:45-46, against the relevance the reference's own Functions produced on the same weights
    (tests/golden/llama_*.npz: head dims 16 / 32 / 128, GQA 2:1 / 4:1) -- < 1e-4 (instances the reference itself resolves in fp32: cond_gap < 5e-6).
    remove() restores the plain model; cp_lrp (ref :95-105) puts no relevance on q / k and conserves it through V."""
    _need_gpu()
    from tests.golden.hf_models import build_llama_from_weights
    from tests.util import llama_case
    from lxt_amd.explicit.models import llama as xl
    cfg, W, ids, fx = llama_case(name)
    model = build_llama_from_weights(cfg, W, attn=impl).cuda()
    with torch.no_grad():
        plain = model(input_ids=ids[None].cuda(), use_cache=False).logits[0, -1].clone()
    xl.attnlrp.register(model)
    try:
        e = model.get_input_embeddings()(ids[None].cuda()).detach().requires_grad_()
        last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
        idx = int(last.argmax())
        assert idx == int(fx["idx"]) and abs(float(last[idx]) - float(fx["logit"])) < 1e-4 and nmax(last.detach(), plain) < 1e-5
        last[idx].backward(last[idx].detach())
        R = e.grad[0].sum(-1)
    finally:
        xl.attnlrp.remove()
    e64, e32 = nmax(R, fx["exp64_R_tok"]), nmax(R, fx["exp32_R_tok"])
    print(f"[explicit Llama composite on HF / {name} / {impl}] token vs reference fp64 {e64:.2e} | vs reference fp32 {e32:.2e} "
          f"(reference's own fp32 gap {float(fx['cond_gap']):.1e}); sum R {float(R.sum()):+.6f} vs {float(t(fx['exp64_R_tok']).sum()):+.6f}")
    assert e64 < max(1e-4, 3 * float(fx["cond_gap"]))
    with torch.no_grad():
        again = model(input_ids=ids[None].cuda(), use_cache=False).logits[0, -1]
    assert torch.equal(again, plain), "remove() must restore the plain model"
    if name == "mid" and impl == "eager":
        xl.cp_lrp.register(model)
        try:
            e = model.get_input_embeddings()(ids[None].cuda()).detach().requires_grad_()
            last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
            last[idx].backward(last[idx].detach())
            Rcp = e.grad[0].sum(-1)
        finally:
            xl.cp_lrp.remove()
        assert torch.isfinite(Rcp).all() and nmax(Rcp, R) > 1e-3          # a different rule set, a different explanation


def test_dropin_fused_mlp_and_hip_rope_match_the_unfused_dropin():
    """The bf16 drop-in path of an adopted decoder runs its gated MLP on the fused-epilogue GEMMs (patches._fused_mlp_weights / FusedGatedMLPFn) and
    HF's apply_rotary_pos_emb on the HIP RoPE kernels (patches.patch_rotary): same explanation as the un-fused drop-in (three GEMMs + rule kernel,
    HF's eager RoPE) up to bf16 rounding of the one differently-ordered sum (gate / up dgrads), and both agree with the fp32 drop-in; fp32 never
    takes the fused path (bit-identical to before)."""
    _need_gpu()
    from transformers.models.llama import modeling_llama
    from lxt_amd.efficient import monkey_patch
    import lxt_amd.efficient.patches as P
    from oracle import llama as ol
    from tests.golden.hf_models import build_llama_from_weights
    cfg = dict(hidden=1024, inter=2816, n_layers=3, n_heads=8, n_kv=2, head_dim=128, vocab=512, rope_theta=5e5, rms_eps=1e-5)
    W = ol.random_weights(cfg, seed=77)
    ids = torch.randint(0, 512, (2, 384), generator=torch.Generator().manual_seed(5)).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkey_patch(modeling_llama)
    hip_rope = modeling_llama.apply_rotary_pos_emb
    assert hip_rope.__module__ == P.__name__

    def run(dtype, fuse, rope):
        model = build_llama_from_weights(cfg, W, attn="sdpa", dtype=dtype).cuda()
        P.FUSE_MLP, modeling_llama.apply_rotary_pos_emb = fuse, (hip_rope if rope else hip_rope.__wrapped__)
        try:
            e = model.get_input_embeddings()(ids).detach().requires_grad_()
            last = model(inputs_embeds=e, use_cache=False).logits[:, -1]
            idx = last.argmax(-1)
            last[torch.arange(2), idx].sum().backward()
            fused_used = any("_lrp_fused_mlp" in m.__dict__ for m in model.modules())
            if fused_used:      # the extra memory can be handed back: the interleaved copies go, down weights return to contiguous storage
                P.release_fused(model)
                assert not any("_lrp_fused_mlp" in m.__dict__ for m in model.modules())
                assert all(m.down_proj.weight.is_contiguous() for m in model.modules() if hasattr(m, "down_proj"))
            return idx.cpu(), (e * e.grad).float().sum(-1).double().cpu(), fused_used
        finally:
            P.FUSE_MLP, modeling_llama.apply_rotary_pos_emb = True, hip_rope
    i32, R32, f32_fused = run(torch.float32, True, True)
    _, R32b, _ = run(torch.float32, False, False)
    assert not f32_fused and nmax(R32, R32b) < 1e-5          # fp32: no fused MLP; the HIP RoPE is the same arithmetic as HF's eager form
    ia, Ra, used = run(torch.bfloat16, True, True)
    ib, Rb, unused = run(torch.bfloat16, False, False)
    assert used and not unused and torch.equal(ia, ib)
    cos = lambda a, b: float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0))   # noqa: E731
    print(f"[drop-in bf16, fused MLP + HIP RoPE vs un-fused] nmax {nmax(Ra, Rb):.2e} cos {cos(Ra, Rb):.6f} | vs fp32 drop-in: fused {nmax(Ra, R32):.2e} "
          f"(cos {cos(Ra, R32):.6f}), un-fused {nmax(Rb, R32):.2e} (cos {cos(Rb, R32):.6f})")
    assert cos(Ra, Rb) > 0.999 and cos(Ra, R32) > 0.995 and nmax(Ra, R32) < 3 * max(nmax(Rb, R32), 1e-2)


def test_dropin_fused_decoder_layer_matches_the_per_module_dropin():
    """Round 6: for an adopted bf16 Llama at M = B S rows the drop-in path runs each decoder layer as ONE autograd node on the engine's fused launch
    sequence (patches.decoder_layer_forward / DecoderLayerFn: fused QKV, K1n norm + residual epilogues, coefficient-stash gated rule, D and RoPE's
    backward inside the attention backward) -- the reference's user protocol (docs/source/quickstart.rst:120-141) unchanged.  Same explanation as
    the per-module drop-in up to bf16 rounding, both against the fp32 drop-in (which never takes the fused layer); a left-padded batch falls back
    to the per-module path and still runs."""
    _need_gpu()
    from transformers.models.llama import modeling_llama
    from lxt_amd.efficient import monkey_patch
    import lxt_amd.efficient.patches as P
    from oracle import llama as ol
    from tests.golden.hf_models import build_llama_from_weights
    cfg = dict(hidden=2048, inter=5632, n_layers=3, n_heads=16, n_kv=4, head_dim=128, vocab=1024, rope_theta=1e4, rms_eps=1e-5)
    W = ol.random_weights(cfg, seed=77)
    g = torch.Generator().manual_seed(78)
    for L in W["layers"]:
        L["ln1"] = (0.25 + 1.5 * torch.rand(cfg["hidden"], generator=g))
        L["ln2"] = (0.25 + 1.5 * torch.rand(cfg["hidden"], generator=g))
    B, S = 3, 2048
    ids = torch.randint(0, cfg["vocab"], (B, S), generator=g).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkey_patch(modeling_llama)
    assert modeling_llama.LlamaDecoderLayer.forward is P.decoder_layer_forward

    def run(dtype, fuse_layer, tgt=None, mask=None):
        model = build_llama_from_weights(cfg, W, attn="sdpa", dtype=dtype, rotary_fp32=True).cuda()      # (inv_freq as from_pretrained keeps it)
        P.FUSE_LAYER = fuse_layer
        try:
            e = model.get_input_embeddings()(ids).detach().requires_grad_()
            last = model(inputs_embeds=e, attention_mask=mask, use_cache=False).logits[:, -1]
            idx = last.argmax(-1) if tgt is None else tgt
            last[torch.arange(B), idx].sum().backward()
            used = [bool(L.__dict__.get("_lrp_fused_layer", {}).get("ok_rows", {}).get((B, S), False)) for L in model.model.layers]
            return idx, (e * e.grad).float().sum(-1).double().cpu(), used
        finally:
            P.FUSE_LAYER = True
    i32, R32, u32 = run(torch.float32, True)
    assert not any(u32)                                               # fp32: HF's forward over the per-module patches, as before
    ia, Ra, ua = run(torch.bfloat16, True, tgt=i32)
    ib, Rb, ub = run(torch.bfloat16, False, tgt=i32)
    assert all(ua) and not any(ub)
    e_a, e_b = [nmax(Ra[b], R32[b]) for b in range(B)], [nmax(Rb[b], R32[b]) for b in range(B)]
    cos = lambda a, b: float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0))   # noqa: E731
    print(f"[drop-in bf16, fused decoder layer vs per-module] vs the fp32 drop-in per prompt: fused layer {[f'{x:.2e}' for x in e_a]} (cos {cos(Ra, R32):.6f}), "
          f"per-module {[f'{x:.2e}' for x in e_b]} (cos {cos(Rb, R32):.6f}); fused vs per-module {nmax(Ra, Rb):.2e}")
    assert torch.isfinite(Ra).all() and cos(Ra, R32) > 0.999 and max(e_a) < max(2e-2, 1.5 * max(e_b))
    # a left-padded batch: the fused layer declines (per-row key intervals), HF's forward over the per-module patches takes over
    mask = torch.ones(B, S, dtype=torch.long, device="cuda")
    mask[0, :100] = 0
    _, Rp, up = run(torch.bfloat16, True, tgt=i32, mask=mask)
    assert torch.isfinite(Rp).all() and float(Rp[0, :100].abs().max()) == 0.0
