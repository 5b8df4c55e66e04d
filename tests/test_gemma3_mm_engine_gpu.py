"""GPU parity of the fused image + text Gemma-3 driver (lxt_amd.engine_gemma3_mm.Gemma3MMLRP: SigLIP tower + projector + text decoder;
BASELINE config 4 "image+text"):
  (1) against the fixture captured from the REAL reference (tests/golden/gemma3_mm.npz: lxt.efficient.monkey_patch(modeling_gemma3) on a
      seeded Gemma3ForConditionalGeneration, fp32 and fp64, relevance of the text tokens AND of the pixels), in both of the reference's
      semantics for the tower's attention (eager: un-patched; sdpa: AttnLRP rule through the process-wide attention registry) -- < 1e-4;
  (2) at the released 4B dimensions (SigLIP H 1152 / 16 heads of d = 72 / I 4304 / 896 x 896 pixels -> 4096 patches -> 256 image tokens, text
      H 2560 / d = 256 / window 1024; two tower and two text layers): fp32 against a second fixture captured from the REAL reference at these
      dimensions (tests/golden/gemma3_mm_4bdims.npz), both attention semantics; bf16 against fp32."""
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
    # ids / token types handed over on the DEVICE (the bookkeeping then takes its one host copy) and the image mask derived from the ids alone:
    # the same explanation, bit for bit
    out_d = eng.explain(ids.cuda(), pv.cuda(), token_type_ids=tt.cuda())
    out_i = eng.explain(ids, pv)
    assert torch.equal(out_d["R_tok"], out["R_tok"]) and torch.equal(out_d["R_pix"], out["R_pix"]) and torch.equal(out_i["R_tok"], out["R_tok"])


@pytest.mark.parametrize("impl", ["sdpa", "eager"])
def test_gemma3_mm_engine_full_dims_vs_reference_fixture(mm, impl):
    """BASELINE config 4 AS NAMED at the released 4B dimensions (SigLIP H 1152 / 16 heads of d = 72 / I 4304 / 896 x 896 pixels -> 4096 patches ->
    256 image tokens, two tower layers; text H 2560 / d = 256 / window 1024, one sliding + one global layer) against the REAL reference:
    tests/golden/gemma3_mm_4bdims.npz holds what `lxt.efficient.monkey_patch(modeling_gemma3)` (ref lxt/efficient/models/gemma3.py:14-19,
    lxt/efficient/patches.py:193-203) produces on the seeded Gemma3ForConditionalGeneration in fp64 on the CPU, for both attention
    implementations (tests/golden/make_golden_gemma3_mm_4bdims.py; the reference's own fp32 is 1.8e-8 / 1.8e-6 / 2.1e-6 from it on token / pixel /
    patch relevance).  Fused fp32 driver < 1e-4 on all three; bf16 driver vs fp32 (sdpa)."""
    from tests.golden.hf_models import build_gemma3_mm_fulldims, gemma3_mm_fulldims_inputs
    import numpy as np
    fx = load("gemma3_mm_4bdims.npz")
    model = build_gemma3_mm_fulldims(attn="sdpa")
    ids, tt, pv = gemma3_mm_fulldims_inputs()
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * abs(float(fx["wsum"])), "weights did not reproduce"
    assert np.array_equal(ids.numpy(), fx["ids"]) and abs(float(pv.double().abs().sum()) - float(fx["pv_sum"])) < 1e-6 * float(fx["pv_sum"])
    eng32 = mm.Gemma3MMLRP.from_hf(model, dtype=torch.float32, max_seq=512, vision_attn_rule=(impl == "sdpa"))
    r32 = eng32.explain(ids, pv, token_type_ids=tt)
    del eng32
    torch.cuda.empty_cache()
    assert int(r32["idx"][0]) == int(fx[f"{impl}_idx"]) and abs(float(r32["logit"][0]) - float(fx[f"{impl}_logit"])) < 1e-3
    rows = t(fx["rows"]).long()
    e_t = nmax(r32["R_tok"][0], fx[f"{impl}_R_tok"])
    e_pa = nmax(r32["R_patch"][0], fx[f"{impl}_R_patch"])
    e_p = float((r32["R_pix"][0].double().cpu()[:, rows] - t(fx[f"{impl}_R_pix_rows"]).double()).abs().max() / float(fx[f"{impl}_R_pix_absmax"]))
    g = fx[f"{impl}_gap"]
    print(f"[gemma3 4B image+text dims / {impl}, fp32 fused driver vs the REFERENCE (fp64)] token {e_t:.2e} | patch {e_pa:.2e} | pixel (64 sampled rows) {e_p:.2e} "
          f"(the reference's own fp32: {g[0]:.1e} | {g[2]:.1e} | {g[1]:.1e}); share of relevance on the image "
          f"{float(r32['R_pix'].sum()) / (float(r32['R_pix'].sum()) + float(r32['R_tok'].sum())):.3f}")
    assert e_t < 1e-4 and e_pa < 1e-4 and e_p < 1e-4
    assert float(r32["R_tok"][0][tt[0].bool().cuda()].abs().max()) == 0.0
    if impl == "sdpa":
        engb = mm.Gemma3MMLRP.from_hf(model, dtype=torch.bfloat16, max_seq=512, vision_attn_rule=True)
        rb = engb.explain(ids, pv, token_type_ids=tt, target=r32["idx"].cpu())
        del engb
        torch.cuda.empty_cache()
        cos_t = float(torch.nn.functional.cosine_similarity(rb["R_tok"][0].double(), r32["R_tok"][0].double(), dim=0))
        cos_p = float(torch.nn.functional.cosine_similarity(rb["R_patch"][0].double().flatten(), r32["R_patch"][0].double().flatten(), dim=0))
        print(f"[gemma3 4B image+text dims, bf16 vs fp32 fused] token nmax {nmax(rb['R_tok'][0], r32['R_tok'][0]):.2e} cos {cos_t:.5f} | patch nmax "
              f"{nmax(rb['R_patch'][0], r32['R_patch'][0]):.2e} cos {cos_p:.5f}")
        assert torch.isfinite(rb["R_pix"]).all() and cos_t > 0.99 and cos_p > 0.98
