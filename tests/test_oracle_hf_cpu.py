"""CPU: the drop-in oracle (oracle/hf_efficient.py) against the fixtures captured from the real
reference on HF BERT-base and a Gemma3 text tower, and against the Llama lxt.efficient fixture."""
import pytest
import torch

from oracle import hf_efficient as oh
from tests.golden.hf_models import build_bert, build_gemma3, wsum
from tests.util import nmax, load, t


def test_bert_oracle_vs_reference():
    fx = load("bert_base.npz")
    model = build_bert(seed=0, attn="eager")
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    out = oh.explain_classifier(oh.patch_instance(model), t(fx["ids"]))
    assert out["idx"] == int(fx["idx"]) and nmax(out["R_tok"], fx["R_tok"]) < 5e-6


def test_gemma3_oracle_vs_reference():
    fx = load("gemma3_tiny.npz")
    model = build_gemma3(seed=3, attn="eager")
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    out = oh.explain_causal_lm(oh.patch_instance(model), t(fx["ids"]))
    assert out["idx"] == int(fx["idx"]) and nmax(out["R_tok"], fx["R_tok"]) < 5e-6


@pytest.mark.parametrize("which", ["llama_cp", "qwen2", "qwen3", "gpt2"])
def test_family_oracle_vs_reference(which):
    from tests.golden.hf_models import BUILDERS
    fx = load(f"hf_{which}.npz")
    model = BUILDERS[which](attn="eager")
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    out = oh.explain_causal_lm(oh.patch_instance(model, "cp" if which.endswith("_cp") else "attnlrp"), t(fx["ids"]))
    assert out["idx"] == int(fx["idx"]) and nmax(out["R_tok"], fx["R_tok"]) < 5e-6


@pytest.mark.parametrize("side", ["left", "right"])
def test_padded_batch_oracle_vs_reference(side):
    """left / right padded batches (HF builds the padding masks): the oracle restatement against the reference fixture"""
    from tests.golden.hf_models import BUILDERS
    fx = load("hf_qwen2_padded.npz")
    model = oh.patch_instance(BUILDERS["qwen2"](attn="eager"))
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    ids, am, pos = t(fx["ids"]), t(fx[f"{side}_mask"]), t(fx[f"{side}_pos"])
    e = model.get_input_embeddings()(ids).detach().requires_grad_()
    logits = model(inputs_embeds=e, attention_mask=am, use_cache=False).logits
    rows = torch.arange(ids.shape[0])
    last = logits[rows, pos]
    idx = last.argmax(-1)
    assert idx.tolist() == t(fx[f"{side}_idx"]).tolist()
    last[rows, idx].sum().backward()
    R = (e * e.grad).sum(-1)
    for b in range(ids.shape[0]):
        valid = am[b].bool()
        assert nmax(R[b][valid], t(fx[f"{side}_R_tok"])[b][valid]) < 5e-6


@pytest.mark.parametrize("impl", ["eager", "sdpa"])
def test_gemma3_image_branch_oracle_vs_reference(impl):
    """Gemma-3 with the SigLIP tower: the two semantics of the reference (eager: SigLIP attention un-patched; sdpa: AttnLRP
    attention rule inside SigLIP too) restated by scoping the oracle's instance patches, against the reference fixture"""
    from tests.golden.hf_models import build_gemma3_mm
    fx = load("gemma3_mm.npz")
    model = build_gemma3_mm(attn="eager")
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * abs(float(fx["wsum"]))
    acfg = ("text_config",) if impl == "eager" else ("text_config", "vision_config")
    model = oh.patch_instance(model, skip=("model.vision_tower", "model.multi_modal_projector.avg_pool"), attn_configs=acfg)
    ids, tt, pv = t(fx["ids"]), t(fx["token_type_ids"]), t(fx["pixel_values"])
    e = model.get_input_embeddings()(ids).detach().requires_grad_()
    px = pv.clone().requires_grad_()
    last = model(inputs_embeds=e, pixel_values=px, token_type_ids=tt, use_cache=False).logits[0, -1]
    idx = int(last.argmax())
    assert idx == int(fx[f"{impl}_idx"])
    last[idx].backward()
    assert nmax((e * e.grad)[0].sum(-1), fx[f"{impl}_R_tok"]) < 5e-6
    assert nmax((px * px.grad)[0], fx[f"{impl}_R_pix"]) < 5e-6


def test_bert_explicit_oracle_vs_reference_fixture():
    """BASELINE config 2 in lxt.explicit semantics: oracle/bert.py (gradient form, fp64) against the per-token / per-neuron
    relevance the reference's own explicit Functions produced (tests/golden/make_golden_bert_explicit.py, BERT-base S = 128)"""
    import numpy as np
    import torch
    from oracle import bert as ob
    from tests.golden import bert_explicit_compose as C
    from tests.golden.hf_models import build_bert, wsum
    from tests.util import load, nmax, t
    fx = load("bert_base_explicit.npz")
    model = build_bert(seed=0, attn="eager")
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    W = C.weights_from_hf(model, torch.float64)
    o = ob.explain(W, t(fx["ids"]), dtype=torch.float64)
    assert o["idx"] == int(fx["idx"]) and abs(o["logit"] - float(fx["logit"])) < 1e-9
    assert nmax(o["R_tok"], fx["R_tok_fp64"]) < 1e-9 and nmax(o["R_emb"], fx["R_emb_fp64"]) < 1e-6      # (R_emb stored as fp32)
    o32 = ob.explain(W, t(fx["ids"]), target=o["idx"], dtype=torch.float32)
    gap = nmax(o32["R_tok"], o["R_tok"])
    print(f"oracle fp32 vs fp64 {gap:.2e} (reference's own gap {float(fx['cond_gap']):.2e})")
    assert gap < 10 * float(fx["cond_gap"])


def test_small_case_reference_fixture_matches_live_oracle():
    """tests/golden/small_cases_ref.npz (the yardsticks of the small explicit-mode GPU cases, written from the imported reference by
    tests/golden/make_golden_small_cases.py): its `R_tok` is the exact result the live fp64 oracle computes on the regenerated instance, and
    the oracle run in fp32 -- the same op sequence as the reference's Functions -- lands where the reference's own fp32 run did."""
    import torch
    from oracle import llama as ol
    from tests.util import ref_case, nmax
    cfg = dict(hidden=256, inter=512, n_layers=2, n_heads=8, n_kv=2, head_dim=32, vocab=512, rope_theta=5e5, rms_eps=1e-5)
    W = ol.random_weights(cfg, seed=302)
    for S, b in ((37, 0), (100, 2)):
        ids = torch.randint(0, 512, (3 if S == 100 else 1, S), generator=torch.Generator().manual_seed(S))[b]
        fx = ref_case(f"llama_ragged_S{S}_b{b}")
        o64 = ol.explain(cfg, W, ids=ids, mode="explicit", dtype=torch.float64)
        o32 = ol.explain(cfg, W, ids=ids, target=o64["idx"], mode="explicit", dtype=torch.float32)
        assert o64["idx"] == fx["idx"] and nmax(o64["R_tok"], fx["R_tok"]) < 1e-10
        gap = nmax(o32["R_tok"], o64["R_tok"])
        print(f"S={S} prompt {b}: oracle fp32 vs exact {gap:.2e}; the reference's own fp32 {fx['gap']:.2e}")
        assert gap < 10 * fx["gap"] + 1e-6
