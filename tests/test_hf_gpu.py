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


@pytest.mark.parametrize("which", ["llama_cp", "qwen2", "qwen3", "gpt2", "qwen2_padded", "gemma3_mm", "mini_vit"])
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
