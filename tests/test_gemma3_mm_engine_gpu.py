"""GPU parity of the fused image + text Gemma-3 driver (lxt_amd.engine_gemma3_mm.Gemma3MMLRP: SigLIP tower + projector + text decoder;
BASELINE config 4 "image+text"):
  (1) against the fixture captured from the REAL reference (tests/golden/gemma3_mm.npz: lxt.efficient.monkey_patch(modeling_gemma3) on a
      seeded Gemma3ForConditionalGeneration, fp32 and fp64, relevance of the text tokens AND of the pixels), in both of the reference's
      semantics for the tower's attention (eager: un-patched; sdpa: AttnLRP rule through the process-wide attention registry) -- < 1e-4;
  (2) at the released 4B dimensions (SigLIP H 1152 / 16 heads of d = 72 / I 4304 / 896 x 896 pixels -> 4096 patches -> 256 image tokens, text
      H 2560 / d = 256 / window 1024; two tower and two text layers): fp32 against the drop-in path (the HF model under
      lxt_amd.efficient.monkey_patch, autograd-driven -- the path the fixture of (1) pins), bf16 against fp32."""
import warnings

import pytest
import torch

from tests.golden.hf_models import build_gemma3_mm, wsum
from tests.util import nmax, load, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mm():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import lxt_amd.engine_gemma3_mm as e
    return e


@pytest.mark.parametrize("impl", ["eager", "sdpa"])
def test_gemma3_mm_engine_fp32_vs_reference_fixture(mm, impl):
    fx = load("gemma3_mm.npz")
    model = build_gemma3_mm(attn="eager")
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * abs(float(fx["wsum"])), "weights did not reproduce"
    eng = mm.Gemma3MMLRP.from_hf(model, dtype=torch.float32, max_seq=256, vision_attn_rule=(impl == "sdpa"))
    ids, tt, pv = t(fx["ids"]), t(fx["token_type_ids"]), t(fx["pixel_values"])
    out = eng.explain(ids, pv, token_type_ids=tt)
    assert int(out["idx"][0]) == int(fx[f"{impl}_idx"]) and abs(float(out["logit"][0]) - float(fx[f"{impl}_logit"])) < 1e-4
    Rt, Rp = out["R_tok"][0], out["R_pix"][0]
    errs = [nmax(Rt, fx[f"{impl}_R_tok"]), nmax(Rt, fx[f"{impl}_R_tok_fp64"]), nmax(Rp, fx[f"{impl}_R_pix"]), nmax(Rp, fx[f"{impl}_R_pix_fp64"])]
    ref_patch = t(fx[f"{impl}_R_pix_fp64"]).reshape(3, 4, 14, 4, 14).sum((0, 2, 4))
    errs.append(nmax(out["R_patch"][0], ref_patch))
    print(f"[gemma3_mm fused / {impl}] text vs ref {errs[0]:.2e} / fp64 {errs[1]:.2e} | pixels vs ref {errs[2]:.2e} / fp64 {errs[3]:.2e} | patches {errs[4]:.2e}; "
          f"sum R text {float(Rt.sum()):+.6f} (ref {float(fx[f'{impl}_R_tok'].sum()):+.6f}) image {float(Rp.sum()):+.6f} (ref {float(fx[f'{impl}_R_pix'].sum()):+.6f})")
    assert max(errs) < 1e-4
    # the word embeddings at image positions were replaced by the image features: exactly zero relevance there
    assert float(Rt[tt[0].bool().cuda()].abs().max()) == 0.0


def test_gemma3_mm_engine_full_dims(mm):
    from tests.golden.hf_models import build_gemma3_mm_fulldims, gemma3_mm_fulldims_inputs
    from transformers.models.gemma3 import modeling_gemma3
    from lxt_amd.efficient import monkey_patch
    model = build_gemma3_mm_fulldims(attn="sdpa")
    ids, tt, pv = gemma3_mm_fulldims_inputs()
    eng32 = mm.Gemma3MMLRP.from_hf(model, dtype=torch.float32, max_seq=512, vision_attn_rule=True)
    r32 = eng32.explain(ids, pv, token_type_ids=tt)
    del eng32
    engb = mm.Gemma3MMLRP.from_hf(model, dtype=torch.bfloat16, max_seq=512, vision_attn_rule=True)
    rb = engb.explain(ids, pv, token_type_ids=tt, target=r32["idx"].cpu())
    del engb
    torch.cuda.empty_cache()
    cos_t = float(torch.nn.functional.cosine_similarity(rb["R_tok"][0].double(), r32["R_tok"][0].double(), dim=0))
    cos_p = float(torch.nn.functional.cosine_similarity(rb["R_patch"][0].double().flatten(), r32["R_patch"][0].double().flatten(), dim=0))
    print(f"[gemma3 4B image+text dims, bf16 vs fp32 fused] token nmax {nmax(rb['R_tok'][0], r32['R_tok'][0]):.2e} cos {cos_t:.5f} | patch nmax "
          f"{nmax(rb['R_patch'][0], r32['R_patch'][0]):.2e} cos {cos_p:.5f}")
    assert torch.isfinite(rb["R_pix"]).all() and cos_t > 0.99 and cos_p > 0.98
    # the drop-in path (class-level patches, autograd) on the same fp32 weights
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkey_patch(modeling_gemma3)
    for p_ in model.parameters():
        p_.requires_grad_(False)
    model = model.cuda()
    e = model.get_input_embeddings()(ids.cuda()).detach().requires_grad_()
    px = pv.cuda().clone().requires_grad_()
    last = model(inputs_embeds=e, pixel_values=px, token_type_ids=tt.cuda(), use_cache=False).logits[0, -1]
    idx = int(r32["idx"][0])
    assert int(last.argmax()) == idx and abs(float(last[idx]) - float(r32["logit"][0])) < 1e-3
    last[idx].backward()
    Rt, Rp = (e * e.grad)[0].sum(-1), (px * px.grad)[0]
    Rpatch = Rp.reshape(3, 64, 14, 64, 14).sum((0, 2, 4))
    e_t, e_p, e_pa = nmax(r32["R_tok"][0], Rt), nmax(r32["R_pix"][0], Rp), nmax(r32["R_patch"][0], Rpatch)
    print(f"[gemma3 4B image+text dims, fp32 fused vs drop-in path] token {e_t:.2e} | pixel {e_p:.2e} | patch {e_pa:.2e}; share of relevance on the image "
          f"{float(r32['R_pix'].sum()) / (float(r32['R_pix'].sum()) + float(r32['R_tok'].sum())):.3f}")
    assert e_t < 1e-4 and e_pa < 1e-4 and e_p < 1e-3
