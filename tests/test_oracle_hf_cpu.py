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
