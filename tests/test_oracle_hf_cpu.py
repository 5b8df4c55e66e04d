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
