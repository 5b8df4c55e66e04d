#!/usr/bin/env python3
"""Fixture generator (build container only; imports the REAL reference from /root/reference):
gemma3_4bdims.npz = `lxt.efficient.monkey_patch(modeling_gemma3)` (ref lxt/efficient/models/gemma3.py:11-19) on a seeded
Gemma3ForCausalLM at the released Gemma-3-4B TEXT-tower layer dimensions (H 2560, 8 query / 4 kv heads of d = 256, I 10240, sliding
window 1024, tied embeddings; tests/golden/hf_models.build_gemma3_4bdims), one local + one global layer, S = 2048 (> the window), run
in fp64 and in fp32 on the CPU.  Protocol: docs/source/quickstart.rst:120-141 (inputs_embeds.requires_grad_(), logits[0,-1,argmax]
.backward(), (e * e.grad).sum(-1)).  Pins BASELINE config 4's text path at full width."""
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

from tests.golden.hf_models import build_gemma3_4bdims, wsum  # noqa: E402


def explain(model, ids, target=None):
    for p in model.parameters():
        p.requires_grad_(False)
    e = model.get_input_embeddings()(ids[None]).detach().requires_grad_()
    last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
    idx = int(last.argmax()) if target is None else target
    last[idx].backward()
    return idx, float(last[idx]), (e * e.grad)[0].sum(-1).detach()


def main():
    from lxt.efficient import monkey_patch
    from transformers.models.gemma3 import modeling_gemma3
    S = 2048
    ids = torch.randint(0, 4096, (S,), generator=torch.Generator().manual_seed(9))
    ws = wsum(build_gemma3_4bdims())
    monkey_patch(modeling_gemma3)
    t0 = time.time()
    idx, logit, R64 = explain(build_gemma3_4bdims().double(), ids)
    print(f"reference fp64: idx {idx} logit {logit:+.8f} sum R {float(R64.sum()):+.8f}  ({time.time() - t0:.0f} s)", flush=True)
    idx32, logit32, R32 = explain(build_gemma3_4bdims(), ids, target=idx)
    gap = float((R32.double() - R64).abs().max() / R64.abs().max())
    print(f"reference fp32: logit {logit32:+.8f}; fp32-vs-fp64 gap {gap:.2e}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "gemma3_4bdims.npz"), ids=ids.numpy(), idx=idx, logit=logit, R_tok_fp64=R64.numpy(),
                        R_tok_fp32=R32.numpy(), ref_fp32_gap=gap, wsum=ws, seed=5, S=S)


if __name__ == "__main__":
    main()
