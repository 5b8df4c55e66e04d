#!/usr/bin/env python3
"""Fixture of BASELINE config 4 AS NAMED (Gemma-3-4B-it image + text) at the released layer dimensions, FROM THE REAL REFERENCE
(VERDICT r4 "what's weak" 3: the 4B-dim image + text test compared the fused driver with the repo's own drop-in path only).

Run in the build container only (needs /root/reference):

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_golden_gemma3_mm_4bdims.py

`lxt.efficient.monkey_patch(modeling_gemma3)` (ref lxt/efficient/models/gemma3.py:14-19, lxt/efficient/patches.py:193-203) is applied to a
seeded `Gemma3ForConditionalGeneration` at the 4B dimensions (tests/golden/hf_models.build_gemma3_mm_fulldims: SigLIP H 1152 / 16 heads of
d = 72 / I 4304 / 896 x 896 pixels -> 4096 patches -> 256 image tokens, two tower layers; text H 2560 / 8 + 4 heads of d = 256 / I 10240 /
window 1024, one sliding + one global layer) and the user protocol of docs/source/quickstart.rst:120-141 is run on the CPU in fp32 and in
fp64 for both attention implementations (sdpa: the tower's attention takes the AttnLRP rule through the process-wide registry; eager: it
does not).  Stored per implementation: explained index, logit, token relevance [S], patch relevance [64, 64] (sum over a 14 x 14 x 3 patch),
pixel relevance on 64 sampled image rows (one per patch row; the full 3 x 896 x 896 tensor would be 10 MB) with the full tensor's |max|.
Inputs and weights are regenerated from seeds on the GPU box and checked through checksums."""
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from tests.golden.hf_models import build_gemma3_mm_fulldims, gemma3_mm_fulldims_inputs, wsum   # noqa: E402

ROWS = np.arange(64) * 14 + 5          # one pixel row inside every patch row


def nmax(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def explain(model, ids, tt, pv, target=None):
    e = model.get_input_embeddings()(ids).detach().requires_grad_()
    p_ = pv.clone().to(e.dtype).requires_grad_()
    last = model(inputs_embeds=e, pixel_values=p_, token_type_ids=tt, use_cache=False).logits[0, -1]
    idx = int(last.argmax()) if target is None else target
    last[idx].backward()
    return idx, float(last[idx]), (e * e.grad)[0].sum(-1).detach(), (p_ * p_.grad)[0].detach()


def main():
    from lxt.efficient import monkey_patch
    from transformers.models.gemma3 import modeling_gemma3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        monkey_patch(modeling_gemma3)
    ids, tt, pv = gemma3_mm_fulldims_inputs()
    save = dict(ids=ids.numpy(), token_type_ids=tt.numpy(), pv_sum=float(pv.double().abs().sum()), rows=ROWS,
                wsum=wsum(build_gemma3_mm_fulldims(attn="sdpa")))
    for impl in ("sdpa", "eager"):
        t0 = time.time()
        model = build_gemma3_mm_fulldims(attn=impl)
        for p_ in model.parameters():
            p_.requires_grad_(False)
        idx, logit, Rt, Rp = explain(model, ids, tt, pv)
        t1 = time.time()
        model = model.double()
        idx64, logit64, Rt64, Rp64 = explain(model, ids, tt, pv.double(), target=idx)
        del model
        Rpa, Rpa64 = Rp.reshape(3, 64, 14, 64, 14).sum((0, 2, 4)), Rp64.reshape(3, 64, 14, 64, 14).sum((0, 2, 4))
        print(f"[gemma3 4B dims image+text / {impl}] idx {idx} logit {logit:+.6f} (fp64 {logit64:+.6f}); sum R text {float(Rt.sum()):+.6f} image "
              f"{float(Rp.sum()):+.6f}; the reference's own fp32 vs fp64: token {nmax(Rt, Rt64):.2e} pixel {nmax(Rp, Rp64):.2e} patch "
              f"{nmax(Rpa, Rpa64):.2e}; fp32 {t1 - t0:.0f} s, fp64 {time.time() - t1:.0f} s", flush=True)
        save.update({f"{impl}_idx": idx, f"{impl}_logit": logit64, f"{impl}_logit32": logit,
                     f"{impl}_R_tok": Rt64.numpy(), f"{impl}_R_patch": Rpa64.numpy(), f"{impl}_R_pix_rows": Rp64[:, ROWS].float().numpy(),
                     f"{impl}_R_pix_absmax": float(Rp64.abs().max()),
                     f"{impl}_gap": np.array([nmax(Rt, Rt64), nmax(Rp, Rp64), nmax(Rpa, Rpa64)])})
    np.savez_compressed(os.path.join(HERE, "gemma3_mm_4bdims.npz"), **save)
    print("wrote gemma3_mm_4bdims.npz")


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
