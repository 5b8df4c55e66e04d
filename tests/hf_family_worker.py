"""Worker (one process per model family -- the patches are class-level and process-global, like the
reference's): patch a HF modeling module with lxt_amd, run the quickstart protocol on the GPU and print
the normalised max error against the fixture captured from the real reference."""
import importlib
import os
import sys
import warnings

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")
from tests.golden.hf_models import BUILDERS, wsum  # noqa: E402
from tests.util import load, t, nmax  # noqa: E402


def padded(which):
    """left / right padded batches: HF's padding masks reach the fused attention as per-row key intervals"""
    fam = which[:-7]
    fx = load(f"hf_{which}.npz")
    mod = importlib.import_module(f"transformers.models.{fam}.modeling_{fam}")
    from lxt_amd.efficient import monkey_patch
    monkey_patch(mod)
    ids = t(fx["ids"]).cuda()
    rows = torch.arange(ids.shape[0], device="cuda")
    worst = 0.0
    for impl in ("eager", "sdpa"):
        model = BUILDERS[fam](attn=impl)
        assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"]), "weights did not reproduce"
        for p in model.parameters():
            p.requires_grad_(False)
        model = model.cuda()
        for side in ("left", "right"):
            am, pos = t(fx[f"{side}_mask"]).cuda(), t(fx[f"{side}_pos"]).cuda()
            e = model.get_input_embeddings()(ids).detach().requires_grad_()
            logits = model(inputs_embeds=e, attention_mask=am, use_cache=False).logits
            last = logits[rows, pos]
            idx = last.argmax(-1)
            assert idx.tolist() == t(fx[f"{side}_idx"]).tolist(), (idx.tolist(), fx[f"{side}_idx"])
            last[rows, idx].sum().backward()
            R = (e * e.grad).sum(-1)
            assert torch.isfinite(R).all()
            valid = am.bool()
            for b in range(ids.shape[0]):
                e32 = nmax(R[b][valid[b]], t(fx[f"{side}_R_tok"])[b][valid[b].cpu()])
                e64 = nmax(R[b][valid[b]], t(fx[f"{side}_R_tok_fp64"])[b][valid[b].cpu()])
                worst = max(worst, e32, e64)
            print(f"[{which}/{impl}/{side}] worst so far {worst:.2e}")
    print(f"WORST {worst:.3e}")
    return 0 if worst < 1e-4 else 1


def gemma3_mm():
    """Gemma-3 with the image branch: relevance of the text tokens and of the pixels against the
    reference's two semantics (eager: SigLIP attention un-patched; sdpa: AttnLRP attention rule inside SigLIP too)"""
    from transformers.models.gemma3 import modeling_gemma3
    from lxt_amd.efficient import monkey_patch
    from tests.golden.hf_models import build_gemma3_mm
    fx = load("gemma3_mm.npz")
    monkey_patch(modeling_gemma3)
    ids, tt, pv = t(fx["ids"]).cuda(), t(fx["token_type_ids"]).cuda(), t(fx["pixel_values"]).cuda()
    worst = 0.0
    for impl in ("eager", "sdpa"):
        model = build_gemma3_mm(attn=impl)
        assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * abs(float(fx["wsum"])), "weights did not reproduce"
        for p in model.parameters():
            p.requires_grad_(False)
        model = model.cuda()
        e = model.get_input_embeddings()(ids).detach().requires_grad_()
        px = pv.clone().requires_grad_()
        last = model(inputs_embeds=e, pixel_values=px, token_type_ids=tt, use_cache=False).logits[0, -1]
        idx = int(last.argmax())
        assert idx == int(fx[f"{impl}_idx"]), (idx, int(fx[f"{impl}_idx"]))
        last[idx].backward()
        Rt, Rp = (e * e.grad)[0].sum(-1), (px * px.grad)[0]
        errs = [nmax(Rt, fx[f"{impl}_R_tok"]), nmax(Rt, fx[f"{impl}_R_tok_fp64"]), nmax(Rp, fx[f"{impl}_R_pix"]), nmax(Rp, fx[f"{impl}_R_pix_fp64"])]
        patch = Rp.reshape(3, 4, 14, 4, 14).sum((0, 2, 4))                      # relevance per ViT patch
        ref_patch = t(fx[f"{impl}_R_pix"]).reshape(3, 4, 14, 4, 14).sum((0, 2, 4))
        errs.append(nmax(patch, ref_patch))
        print(f"[gemma3_mm/{impl}] text vs ref {errs[0]:.2e} / fp64 {errs[1]:.2e} | pixels vs ref {errs[2]:.2e} / fp64 {errs[3]:.2e} | patches {errs[4]:.2e}")
        worst = max(worst, *errs)
    print(f"WORST {worst:.3e}")
    return 0 if worst < 1e-4 else 1


def gemma3_mm_4bdims():
    """the drop-in path (HF Gemma3ForConditionalGeneration under lxt_amd.efficient.monkey_patch, autograd-driven) at the released 4B dimensions
    against the fixture captured from the REAL reference at those dimensions (tests/golden/gemma3_mm_4bdims.npz), both attention semantics"""
    import numpy as np
    from transformers.models.gemma3 import modeling_gemma3
    from lxt_amd.efficient import monkey_patch
    from tests.golden.hf_models import build_gemma3_mm_fulldims, gemma3_mm_fulldims_inputs
    fx = load("gemma3_mm_4bdims.npz")
    monkey_patch(modeling_gemma3)
    ids, tt, pv = gemma3_mm_fulldims_inputs()
    assert np.array_equal(ids.numpy(), fx["ids"])
    rows = t(fx["rows"]).long()
    worst = 0.0
    for impl in ("sdpa", "eager"):
        model = build_gemma3_mm_fulldims(attn=impl)
        assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * abs(float(fx["wsum"])), "weights did not reproduce"
        for p in model.parameters():
            p.requires_grad_(False)
        model = model.cuda()
        e = model.get_input_embeddings()(ids.cuda()).detach().requires_grad_()
        px = pv.cuda().clone().requires_grad_()
        last = model(inputs_embeds=e, pixel_values=px, token_type_ids=tt.cuda(), use_cache=False).logits[0, -1]
        idx = int(last.argmax())
        assert idx == int(fx[f"{impl}_idx"]), (idx, int(fx[f"{impl}_idx"]))
        last[idx].backward()
        Rt, Rp = (e * e.grad)[0].sum(-1), (px * px.grad)[0]
        patch = Rp.reshape(3, 64, 14, 64, 14).sum((0, 2, 4))
        errs = [nmax(Rt, fx[f"{impl}_R_tok"]), nmax(patch, fx[f"{impl}_R_patch"]),
                float((Rp.double().cpu()[:, rows] - t(fx[f"{impl}_R_pix_rows"]).double()).abs().max() / float(fx[f"{impl}_R_pix_absmax"]))]
        print(f"[gemma3_mm 4B dims / {impl}, drop-in fp32 vs the REFERENCE fp64] token {errs[0]:.2e} | patch {errs[1]:.2e} | pixel rows {errs[2]:.2e}")
        worst = max(worst, *errs)
        del model
        torch.cuda.empty_cache()
    print(f"WORST {worst:.3e}")
    return 0 if worst < 1e-4 else 1


def mini_vit():
    """ViT from torch.nn classes under lxt_amd's vit_torch cp_LRP map (explicit patch_map: torchvision is absent) against
    the pixel relevance captured from the reference's patches; the attention runs on the HIP CP path (only dV)"""
    import types
    from lxt_amd.efficient import monkey_patch
    from lxt_amd.efficient.models.vit_torch import cp_LRP
    from tests.golden.hf_models import build_mini_vit
    fx = load("mini_vit.npz")
    monkey_patch(types.ModuleType("mini_vit"), cp_LRP)
    model = build_mini_vit()
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * abs(float(fx["wsum"])), "weights did not reproduce"
    from lxt_amd.efficient import adopt
    model = adopt(model.cuda())          # a plain torch.nn model: its Linear / LayerNorm / Conv2d instances join the HIP path
    x = t(fx["x"]).cuda().requires_grad_()
    y = model(x)
    idx = y.argmax(-1)
    assert idx.tolist() == t(fx["idx"]).tolist()
    y[torch.arange(2, device="cuda"), idx].sum().backward()
    R = x * x.grad
    errs = [nmax(y, fx["logits"]), nmax(R, fx["R_pix"]), nmax(R, fx["R_pix_fp64"]), nmax(R.sum(1), t(fx["R_pix"]).sum(1))]
    print(f"[mini_vit] logits {errs[0]:.2e} | pixel relevance vs ref {errs[1]:.2e} / fp64 {errs[2]:.2e} | heat-map (sum over channels) {errs[3]:.2e}")
    worst = max(errs)
    print(f"WORST {worst:.3e}")
    return 0 if worst < 1e-4 else 1


def bert_explicit():
    """unmodified HF BertForSequenceClassification re-wired in place by lxt_amd.explicit.models.bert.attnlrp (explicit protocol:
    seed the logit with its value, relevance = inputs_embeds.grad) against the reference's own Functions (bert_base_explicit.npz)"""
    from lxt_amd.explicit.models import bert as xb
    from tests.golden.hf_models import build_bert
    fx = load("bert_base_explicit.npz")
    ids = t(fx["ids"])
    model = build_bert(seed=0, attn="eager")
    assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"])
    model = model.cuda()
    xb.attnlrp.register(model)
    e = model.get_input_embeddings()(ids[None].cuda()).detach().requires_grad_()
    logits = model(inputs_embeds=e).logits
    idx = int(logits[0].argmax())
    assert idx == int(fx["idx"]) and abs(float(logits[0, idx]) - float(fx["logit"])) < 1e-4
    logits[0, idx].backward(logits[0, idx].detach())
    R = e.grad[0].sum(-1)
    err = nmax(R, fx["R_tok_fp64"])
    print(f"[bert-base explicit / HF instance + attnlrp.register] token vs reference fp64 {err:.2e} | vs reference fp32 {nmax(R, fx['R_tok']):.2e} "
          f"| sum R {float(R.sum()):.6f} (reference {float(t(fx['R_tok_fp64']).sum()):.6f}; reference's own fp32 gap {float(fx['cond_gap']):.1e})")
    xb.attnlrp.remove()
    with torch.no_grad():
        plain = model(input_ids=ids[None].cuda()).logits[0]
    ok = err < max(1e-4, 3 * float(fx["cond_gap"])) and nmax(plain, fx["logits"]) < 1e-5       # remove() restores the plain model
    print(f"WORST {err:.3e}")
    return 0 if ok else 1


def bert_explicit_padded():
    """a RIGHT-PADDED batch through attnlrp.register (without a mask function registered for the custom attention name HF
    hands attention_mask=None to it and pad tokens are attended to): logits equal HF's eager logits under the same mask, pad
    positions carry exactly zero relevance, and each row equals the fp64 oracle on the un-padded prompt within 1e-4 or 3x what the
    REFERENCE's own explicit composite loses in fp32 on that prompt (tests/golden/small_cases_ref.npz: lxt.explicit.functional / rules composed as
    lxt/explicit/models/bert.py, run in the build container -- explicit stabilisers have poles)."""
    from lxt_amd.explicit.models import bert as xb
    from oracle import bert as ob
    from tests.golden import bert_explicit_compose as C
    from tests.golden.hf_models import build_bert
    from tests.util import bert_oracle, ref_case, ref_bar
    model = build_bert(seed=0, attn="eager")
    W64 = C.weights_from_hf(model, torch.float64)
    model = model.cuda()
    S, lens = 128, (128, 100)
    ids = torch.randint(0, model.config.vocab_size, (2, S), generator=torch.Generator().manual_seed(11))
    am = torch.zeros(2, S, dtype=torch.long)
    for b, L in enumerate(lens):
        am[b, :L] = 1
    with torch.no_grad():
        plain = model(input_ids=ids.cuda(), attention_mask=am.cuda()).logits
        nomask = model(input_ids=ids.cuda()).logits
    xb.attnlrp.register(model)
    e = model.get_input_embeddings()(ids.cuda()).detach().requires_grad_()
    logits = model(inputs_embeds=e, attention_mask=am.cuda()).logits
    idx = logits.argmax(-1)
    sel = logits.gather(1, idx[:, None])[:, 0]
    sel.backward(sel.detach())
    R = e.grad.sum(-1).double().cpu()
    xb.attnlrp.remove()
    ok = nmax(logits.detach(), plain) < 1e-5 and nmax(nomask[1], plain[1]) > 1e-4        # the mask matters for the padded row
    ok = ok and float(R[1, lens[1]:].abs().max()) == 0.0
    worst = 0.0
    for b, L in enumerate(lens):
        o64 = bert_oracle(W64, ids[b, :L], int(idx[b]), draws=3, rel=1e-7)          # fp64 oracle (cached fixture)
        fx = ref_case(f"bert_padded_b{b}")
        err = nmax(R[b, :L], o64["R_tok"])
        print(f"[bert-base explicit padded batch, row {b}, length {L}] token vs oracle fp64 {err:.2e} (the reference's own fp32 on this prompt "
              f"{fx['gap']:.1e}) logit {float(sel[b]):+.6f} vs oracle {o64['logit']:+.6f}")
        ok = ok and int(idx[b]) == fx["idx"] and nmax(o64["R_tok"], fx["R_tok"]) < 1e-9
        ok = ok and abs(float(sel[b]) - o64["logit"]) < 1e-4 and err < ref_bar(fx["gap"], factor=10.0)
        worst = max(worst, err)
    print(f"WORST {worst:.3e}")
    return 0 if ok else 1


def main(which):
    if which == "bert_explicit":
        return bert_explicit()
    if which == "bert_explicit_padded":
        return bert_explicit_padded()
    if which == "mini_vit":
        return mini_vit()
    if which == "gemma3_mm":
        return gemma3_mm()
    if which == "gemma3_mm_4bdims":
        return gemma3_mm_4bdims()
    if which.endswith("_padded"):
        return padded(which)
    fx = load(f"hf_{which}.npz")
    fam = which.replace("_cp", "")
    mod = importlib.import_module(f"transformers.models.{fam}.modeling_{fam}")
    from lxt_amd.efficient import monkey_patch
    if which.endswith("_cp"):
        monkey_patch(mod, importlib.import_module(f"lxt_amd.efficient.models.{fam}").cp_LRP)
    else:
        monkey_patch(mod)
    ids = t(fx["ids"])
    worst = 0.0
    for impl in ("eager", "sdpa"):
        model = BUILDERS[which](attn=impl)
        assert abs(wsum(model) - float(fx["wsum"])) < 1e-6 * float(fx["wsum"]), "weights did not reproduce"
        for p in model.parameters():
            p.requires_grad_(False)
        model = model.cuda()
        e = model.get_input_embeddings()(ids[None].cuda()).requires_grad_()
        last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
        idx = int(last.argmax())
        assert idx == int(fx["idx"]), (idx, int(fx["idx"]))
        last[idx].backward()
        R = (e * e.grad)[0].sum(-1)
        err = max(nmax(R, fx["R_tok"]), nmax(R, fx["R_tok_fp64"]))
        print(f"[{which}/{impl}] tok vs reference {nmax(R, fx['R_tok']):.2e} | vs oracle fp64 {nmax(R, fx['R_tok_fp64']):.2e}")
        worst = max(worst, err)
    print(f"WORST {worst:.3e}")
    return 0 if worst < 1e-4 else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
