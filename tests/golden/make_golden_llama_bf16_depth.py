#!/usr/bin/env python3
"""Fixture generator (build container only; imports the REAL reference from /root/reference):
llama_bf16_depth.npz = `lxt.efficient.monkey_patch(modeling_llama)` (ref lxt/efficient/models/llama.py:9-14) on a seeded, random-init
LlamaForCausalLM at the Llama-3-8B layer dimensions, 8 layers, S = 512, run on the CPU in fp32 AND in bf16 (the reference's own arithmetic
in bf16: HF's bf16 forward, autograd in bf16).  What it pins: how far the REFERENCE ITSELF moves when it is run in bf16 on a deep random-init
model -- the yardstick for the bf16 drop-in path (the same HF forward, the rules on HIP kernels), which the fp32-tolerance tests cannot
give.  Protocol: docs/source/quickstart.rst:120-141."""
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, "/root/reference")
warnings.simplefilter("ignore")

from tests.golden.hf_models import build_llama_8bdims, wsum  # noqa: E402


def explain(model, ids, target=None):
    for p in model.parameters():
        p.requires_grad_(False)
    e = model.get_input_embeddings()(ids[None]).detach().requires_grad_()
    last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
    idx = int(last.argmax()) if target is None else target
    last[idx].backward()
    return idx, float(last[idx]), (e * e.grad)[0].float().sum(-1).detach()


def main():
    from lxt.efficient import monkey_patch
    from transformers.models.llama import modeling_llama
    L, S = 8, 512
    ids = torch.randint(0, 4096, (S,), generator=torch.Generator().manual_seed(21))
    model = build_llama_8bdims(layers=L)
    ws = wsum(model)
    monkey_patch(modeling_llama)
    t0 = time.time()
    idx, logit, R32 = explain(model, ids)
    print(f"reference fp32: idx {idx} logit {logit:+.6f} sum R {float(R32.sum()):+.6f}  ({time.time() - t0:.0f} s)", flush=True)
    mb = model.to(torch.bfloat16)
    _, logit16, R16 = explain(mb, ids, target=idx)
    nm = float((R16.double() - R32.double()).abs().max() / R32.abs().max())
    cos = float(torch.nn.functional.cosine_similarity(R16.double(), R32.double(), dim=0))
    print(f"reference bf16: logit {logit16:+.6f}; vs its own fp32: normalised max err {nm:.2e}, cosine {cos:.5f}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "llama_bf16_depth.npz"), ids=ids.numpy(), idx=idx, logit=logit, logit_bf16=logit16, R_tok_fp32=R32.numpy(),
                        R_tok_bf16=R16.numpy(), ref_bf16_nmax=nm, ref_bf16_cos=cos, wsum=ws, seed=17, S=S, layers=L)


if __name__ == "__main__":
    main()
