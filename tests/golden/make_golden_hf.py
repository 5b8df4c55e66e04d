#!/usr/bin/env python3
"""Golden fixtures for the DROP-IN path on HuggingFace models, generated FROM THE REAL REFERENCE
(build container only):   python tests/golden/make_golden_hf.py

  bert_base.npz    BASELINE config 2: HF BertForSequenceClassification(BertConfig(num_labels=2)),
                   random init (manual_seed 0), S=128, fp32; relevance from the reference's own
                   primitives (lxt.efficient.rules / patches) spliced into HF's modeling_bert with a
                   custom patch_map -- the reference's vendored BERT does not run under transformers 5.x
                   (SURVEY.md finding 9), its marked edit lines are lxt/efficient/models/bert.py:
                   339,380,476-488,581,790,806.
  gemma3_tiny.npz  Gemma3ForCausalLM (text tower, sliding + global layers, q/k-norm, (1+w) RMSNorm,
                   gelu-tanh) through the reference's own default map lxt/efficient/models/gemma3.py.
Each model is built from a seed (weights are not stored; a checksum is), the oracle
oracle/hf_efficient.py is asserted against the reference here.
"""
import os
import sys
import warnings
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")
warnings.simplefilter("ignore")

from oracle import hf_efficient as oh          # noqa: E402
from tests.golden.hf_models import build_bert, build_gemma3, wsum  # noqa: E402


def nmax(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max())


def bert():
    import lxt.efficient.patches as lp
    from lxt.efficient.rules import identity_rule_implicit
    from lxt.efficient import monkey_patch
    from transformers.models.bert import modeling_bert
    from transformers.models.bert.modeling_bert import BertIntermediate, BertPooler
    ids = torch.randint(0, 30522, (128,), generator=torch.Generator().manual_seed(1234))
    # ---- oracle first (instance level, nothing global yet)
    m_or = oh.patch_instance(build_bert(seed=0, attn="eager"))
    o32 = oh.explain_classifier(m_or, ids)
    m64 = oh.patch_instance(build_bert(seed=0, attn="eager").double())
    o64 = oh.explain_classifier(m64, ids, target=o32["idx"])
    # ---- the real reference primitives, custom patch map on HF's own BERT
    def inter_fwd(self, h):
        return identity_rule_implicit(self.intermediate_act_fn, self.dense(h))

    def pool_fwd(self, h):
        return identity_rule_implicit(self.activation, self.dense(h[:, 0]))
    pm = {torch.nn.LayerNorm: partial(lp.patch_method, lp.layer_norm_forward),
          torch.nn.Dropout: partial(lp.patch_method, lp.dropout_forward),
          BertIntermediate: partial(lp.patch_method, inter_fwd), BertPooler: partial(lp.patch_method, pool_fwd),
          modeling_bert: lp.patch_attention}
    monkey_patch(modeling_bert, pm)
    out = {}
    for impl in ("eager", "sdpa"):
        model = build_bert(seed=0, attn=impl)
        for p in model.parameters():
            p.requires_grad_(False)
        e = model.get_input_embeddings()(ids[None]).requires_grad_()
        logits = model(inputs_embeds=e).logits[0]
        idx = int(logits.argmax())
        logits[idx].backward()
        out[impl] = dict(idx=idx, logit=float(logits[idx]), R=(e * e.grad)[0].sum(-1).detach())
    ref = out["eager"]
    print(f"  [bert] idx={ref['idx']} logit={ref['logit']:.6f} sumR={float(ref['R'].sum()):.6f}  eager-vs-sdpa {nmax(out['sdpa']['R'], ref['R']):.2e}")
    print(f"     oracle fp32 vs reference {nmax(o32['R_tok'], ref['R']):.2e} ; oracle fp64 vs reference {nmax(o64['R_tok'], ref['R']):.2e}")
    assert o32["idx"] == ref["idx"] and nmax(o32["R_tok"], ref["R"]) < 2e-5
    np.savez_compressed(os.path.join(HERE, "bert_base.npz"), ids=ids.numpy(), idx=ref["idx"], logit=ref["logit"],
                        R_tok=ref["R"].numpy(), R_tok_fp64=o64["R_tok"].float().numpy(), wsum=wsum(build_bert(seed=0)), seed=0, S=128)


def gemma3():
    from lxt.efficient import monkey_patch
    from transformers.models.gemma3 import modeling_gemma3
    ids = torch.randint(0, 512, (96,), generator=torch.Generator().manual_seed(77))
    m_or = oh.patch_instance(build_gemma3(seed=3, attn="eager"))
    o32 = oh.explain_causal_lm(m_or, ids)
    m64 = oh.patch_instance(build_gemma3(seed=3, attn="eager").double())
    o64 = oh.explain_causal_lm(m64, ids, target=o32["idx"])
    monkey_patch(modeling_gemma3)
    model = build_gemma3(seed=3, attn="eager")
    for p in model.parameters():
        p.requires_grad_(False)
    e = model.get_input_embeddings()(ids[None]).requires_grad_()
    last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
    idx = int(last.argmax())
    last[idx].backward()
    R = (e * e.grad)[0].sum(-1).detach()
    print(f"  [gemma3] idx={idx} logit={float(last[idx]):.6f} sumR={float(R.sum()):.6f}")
    print(f"     oracle fp32 vs reference {nmax(o32['R_tok'], R):.2e} ; oracle fp64 vs reference {nmax(o64['R_tok'], R):.2e}")
    assert o32["idx"] == idx and nmax(o32["R_tok"], R) < 2e-5
    np.savez_compressed(os.path.join(HERE, "gemma3_tiny.npz"), ids=ids.numpy(), idx=idx, logit=float(last[idx]), R_tok=R.numpy(),
                        R_tok_fp64=o64["R_tok"].float().numpy(), wsum=wsum(build_gemma3(seed=3)), seed=3, S=96)


def _explain_mm(model, ids, tt, pv, target=None):
    e = model.get_input_embeddings()(ids).detach().requires_grad_()
    p_ = pv.clone().to(e.dtype).requires_grad_()
    last = model(inputs_embeds=e, pixel_values=p_, token_type_ids=tt, use_cache=False).logits[0, -1]
    idx = int(last.argmax()) if target is None else target
    last[idx].backward()
    return idx, float(last[idx]), (e * e.grad)[0].sum(-1).detach(), (p_ * p_.grad)[0].detach()


def gemma3_mm():
    """Gemma-3 WITH the image branch.  The reference's gemma3 map patches nothing in modeling_siglip, so
    its semantics for the image tower are: plain gradient x input through SigLIP's LayerNorm / GELU / Linear, and the
    AttnLRP attention rule only where the process-wide attention registry is used (sdpa), not with eager.  Both variants
    are captured: relevance of the text tokens AND of the pixels (patch relevance = sum over a 14x14x3 patch)."""
    from lxt.efficient import monkey_patch
    from transformers.models.gemma3 import modeling_gemma3
    from tests.golden.hf_models import build_gemma3_mm, gemma3_mm_inputs
    ids, tt, pv = gemma3_mm_inputs()
    skip = ("model.vision_tower", "model.multi_modal_projector.avg_pool")
    orc = {}
    for impl, acfg in (("eager", ("text_config",)), ("sdpa", ("text_config", "vision_config"))):
        m32 = oh.patch_instance(build_gemma3_mm(attn="eager"), skip=skip, attn_configs=acfg)
        orc[impl, 32] = _explain_mm(m32, ids, tt, pv)
        m64 = oh.patch_instance(build_gemma3_mm(attn="eager").double(), skip=skip, attn_configs=acfg)
        orc[impl, 64] = _explain_mm(m64, ids, tt, pv.double(), target=orc[impl, 32][0])
    monkey_patch(modeling_gemma3)
    save = dict(ids=ids.numpy(), token_type_ids=tt.numpy(), pixel_values=pv.numpy(), wsum=wsum(build_gemma3_mm()))
    for impl in ("eager", "sdpa"):
        model = build_gemma3_mm(attn=impl)
        for p_ in model.parameters():
            p_.requires_grad_(False)
        idx, logit, Rt, Rp = _explain_mm(model, ids, tt, pv)
        o32, o64 = orc[impl, 32], orc[impl, 64]
        print(f"  [gemma3_mm/{impl}] idx={idx} logit={logit:.6f} sumR text={float(Rt.sum()):.6f} image={float(Rp.sum()):.6f}")
        print(f"     oracle fp32 vs reference: text {nmax(o32[2], Rt):.2e} pixels {nmax(o32[3], Rp):.2e} ; "
              f"oracle fp64 vs reference: text {nmax(o64[2], Rt):.2e} pixels {nmax(o64[3], Rp):.2e}")
        assert o32[0] == idx and nmax(o32[2], Rt) < 2e-5 and nmax(o32[3], Rp) < 2e-5
        save.update({f"{impl}_idx": idx, f"{impl}_logit": logit, f"{impl}_R_tok": Rt.numpy(), f"{impl}_R_pix": Rp.numpy(),
                     f"{impl}_R_tok_fp64": o64[2].float().numpy(), f"{impl}_R_pix_fp64": o64[3].float().numpy()})
    np.savez_compressed(os.path.join(HERE, "gemma3_mm.npz"), **save)


def mini_vit():
    """SURVEY 8f rank 3 without torchvision: a ViT built from the torch.nn classes the reference's vit_torch map patches
    (nn.GELU identity rule, nn.LayerNorm, nn.MultiheadAttention CP-LRP: lxt/efficient/models/vit_torch.py:7-11), explained
    with the quickstart protocol (heatmap = (x * x.grad).sum(1)).  zennit's Gamma rule is NOT part of this fixture."""
    from functools import partial
    from torch import nn
    from lxt.efficient import monkey_patch
    from lxt.efficient.patches import patch_method, non_linear_forward, layer_norm_forward, cp_multi_head_attention_forward
    from tests.golden.hf_models import build_mini_vit
    import types
    cp_map = {nn.GELU: partial(patch_method, non_linear_forward, keep_original=True),
              nn.LayerNorm: partial(patch_method, layer_norm_forward),
              nn.MultiheadAttention: partial(patch_method, cp_multi_head_attention_forward, keep_original=True)}
    monkey_patch(types.ModuleType("mini_vit"), cp_map)
    x0 = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(32))
    res = {}
    for dt in (torch.float32, torch.float64):
        model = build_mini_vit().to(dt)
        x = x0.clone().to(dt).requires_grad_()
        y = model(x)
        idx = y.argmax(-1)
        y[torch.arange(2), idx].sum().backward()
        res[dt] = (idx, y.detach(), (x * x.grad).detach())
    gap = nmax(res[torch.float32][2], res[torch.float64][2])
    idx, y, R = res[torch.float32]
    print(f"  [mini_vit] idx={idx.tolist()} logits={[round(float(v), 5) for v in y[torch.arange(2), idx]]} sumR={[round(float(R[b].sum()), 5) for b in range(2)]} "
          f"reference fp32-vs-fp64 {gap:.1e}")
    assert gap < 5e-5
    np.savez_compressed(os.path.join(HERE, "mini_vit.npz"), x=x0.numpy(), idx=idx.numpy(), logits=y.numpy(), R_pix=R.numpy(),
                        R_pix_fp64=res[torch.float64][2].float().numpy(), wsum=wsum(build_mini_vit()), cond_gap=gap)


def family(which):
    """tiny causal LMs through the reference's own default maps: qwen2, qwen3, gpt2 (attnLRP) and llama (cp_LRP)"""
    import importlib
    from lxt.efficient import monkey_patch
    from tests.golden.hf_models import BUILDERS
    build = BUILDERS[which]
    ids = torch.randint(0, 256, (80,), generator=torch.Generator().manual_seed(99))
    variant = "cp" if which.endswith("_cp") else "attnlrp"
    o32 = oh.explain_causal_lm(oh.patch_instance(build(attn="eager"), variant), ids)
    o64 = oh.explain_causal_lm(oh.patch_instance(build(attn="eager").double(), variant), ids, target=o32["idx"])
    fam = which.replace("_cp", "")
    mod = importlib.import_module(f"transformers.models.{fam}.modeling_{fam}")
    if variant == "cp":
        lmap = importlib.import_module(f"lxt.efficient.models.{fam}").cp_LRP
        monkey_patch(mod, lmap)
    else:
        monkey_patch(mod)
    model = build(attn="eager")
    for p in model.parameters():
        p.requires_grad_(False)
    e = model.get_input_embeddings()(ids[None]).requires_grad_()
    last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
    idx = int(last.argmax())
    last[idx].backward()
    R = (e * e.grad)[0].sum(-1).detach()
    print(f"  [{which}] idx={idx} logit={float(last[idx]):.6f} sumR={float(R.sum()):.6f}")
    print(f"     oracle fp32 vs reference {nmax(o32['R_tok'], R):.2e} ; oracle fp64 vs reference {nmax(o64['R_tok'], R):.2e}")
    assert o32["idx"] == idx and nmax(o32["R_tok"], R) < 2e-5
    np.savez_compressed(os.path.join(HERE, f"hf_{which}.npz"), ids=ids.numpy(), idx=idx, logit=float(last[idx]), R_tok=R.numpy(),
                        R_tok_fp64=o64["R_tok"].float().numpy(), wsum=wsum(build()), S=80)


def _explain_padded(model, ids, am, pos):
    """batch of padded prompts: seed every row's arg-max logit at its last REAL position (rows are independent)"""
    e = model.get_input_embeddings()(ids).detach().requires_grad_()
    logits = model(inputs_embeds=e, attention_mask=am, use_cache=False).logits
    rows = torch.arange(ids.shape[0])
    last = logits[rows, pos]
    idx = last.argmax(-1)
    last[rows, idx].sum().backward()
    return idx, last[rows, idx].detach(), (e * e.grad).sum(-1).detach()


def padded(fam="qwen2"):
    """left- and right-padded batches (attention_mask given) through the reference's default map: the masks HF builds
    from a padding mask are per-row key intervals for the fused attention kernel"""
    import importlib
    from lxt.efficient import monkey_patch
    from tests.golden.hf_models import BUILDERS
    build = BUILDERS[fam]
    S, lens = 80, [80, 57, 33]
    g = torch.Generator().manual_seed(4242)
    ids = torch.randint(1, 256, (len(lens), S), generator=g)
    out = {}
    cases = {}
    for side in ("left", "right"):
        am = torch.zeros(len(lens), S, dtype=torch.long)
        for b, n in enumerate(lens):
            if side == "left":
                am[b, S - n:] = 1
            else:
                am[b, :n] = 1
        pos = torch.full((len(lens),), S - 1) if side == "left" else torch.tensor(lens) - 1
        cases[side] = (am, pos)
        m_or = oh.patch_instance(build(attn="eager"))
        out[side, "o32"] = _explain_padded(m_or, ids, am, pos)
        m64 = oh.patch_instance(build(attn="eager").double())
        out[side, "o64"] = _explain_padded(m64, ids, am, pos)
    monkey_patch(importlib.import_module(f"transformers.models.{fam}.modeling_{fam}"))
    save = dict(ids=ids.numpy(), lens=np.array(lens), wsum=wsum(build()), S=S)
    for side, (am, pos) in cases.items():
        model = build(attn="eager")
        for p_ in model.parameters():
            p_.requires_grad_(False)
        idx, logit, R = _explain_padded(model, ids, am, pos)
        o32, o64 = out[side, "o32"], out[side, "o64"]
        valid = am.bool()
        e32 = max(nmax(o32[2][b][valid[b]], R[b][valid[b]]) for b in range(len(lens)))
        e64 = max(nmax(o64[2][b][valid[b]], R[b][valid[b]]) for b in range(len(lens)))
        print(f"  [{fam} {side}-padded] idx={idx.tolist()} sumR={[round(float(R[b][valid[b]].sum()), 5) for b in range(len(lens))]}")
        print(f"     oracle fp32 vs reference {e32:.2e} ; oracle fp64 vs reference {e64:.2e}")
        assert torch.equal(o32[0], idx) and e32 < 2e-5
        save.update({f"{side}_mask": am.numpy(), f"{side}_pos": pos.numpy(), f"{side}_idx": idx.numpy(), f"{side}_logit": logit.numpy(),
                     f"{side}_R_tok": R.numpy(), f"{side}_R_tok_fp64": o64[2].float().numpy()})
    np.savez_compressed(os.path.join(HERE, f"hf_{fam}_padded.npz"), **save)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    # lxt's patches are process-global: one model family per process
    if which == "all":
        import subprocess
        for w in ("bert", "gemma3", "llama_cp", "qwen2", "qwen3", "gpt2", "qwen2_padded", "gemma3_mm", "mini_vit"):
            subprocess.run([sys.executable, os.path.abspath(__file__), w], check=True)
    elif which == "bert":
        bert()
    elif which == "gemma3":
        gemma3()
    elif which == "gemma3_mm":
        gemma3_mm()
    elif which == "mini_vit":
        mini_vit()
    elif which.endswith("_padded"):
        padded(which[:-7])
    else:
        family(which)
