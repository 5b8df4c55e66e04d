#!/usr/bin/env python3
"""BASELINE config 4, text tower: Gemma-3-4B SHAPE (34 layers, H 2560, 8 / 4 heads of d = 256, I 10240, window 1024, 5 local : 1 global,
vocab 262208 tied; random init on the device), S = 2048, bf16.  Fused driver (lxt_amd.engine_gemma3.Gemma3LRP) vs the drop-in path
(the same HF model under lxt_amd.efficient.monkey_patch, autograd-driven) on the same weights.  Dev tool; not the judged benchmark."""
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
warnings.simplefilter("ignore")
from transformers import Gemma3TextConfig, Gemma3ForCausalLM  # noqa: E402
from transformers.models.gemma3 import modeling_gemma3  # noqa: E402
from lxt_amd.efficient import monkey_patch  # noqa: E402
from lxt_amd.engine_gemma3 import Gemma3LRP  # noqa: E402

L = int(os.environ.get("G3_TEXT_LAYERS", 34))
S = int(os.environ.get("G3_SEQ", 2048))
cfg = Gemma3TextConfig(hidden_size=2560, intermediate_size=10240, num_hidden_layers=L, num_attention_heads=8, num_key_value_heads=4,
                       head_dim=256, vocab_size=262208, sliding_window=1024, max_position_embeddings=8192, query_pre_attn_scalar=256,
                       layer_types=[("full_attention" if (i + 1) % 6 == 0 else "sliding_attention") for i in range(L)],
                       attn_implementation="sdpa", tie_word_embeddings=True)
torch.manual_seed(0)
with torch.device("cuda"):
    model = Gemma3ForCausalLM(cfg).to(torch.bfloat16).eval()
for p in model.parameters():
    p.requires_grad_(False)
for n_, p_ in model.named_parameters():
    if "norm" in n_:
        p_.normal_(0, 0.1)
print(f"params {sum(p.numel() for p in model.parameters()) / 1e9:.2f} B, {L} layers, S = {S}", flush=True)
eng = Gemma3LRP.from_hf(model, max_seq=S)
monkey_patch(modeling_gemma3)


def timed(f, n):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for B in (1, 4):
    ids = torch.randint(0, 262000, (B, S), generator=torch.Generator().manual_seed(B)).cuda()
    out = eng.explain(ids)
    dt_e = timed(lambda: eng.explain(ids), 5)

    def dropin():
        R = []
        for b in range(B):
            e = model.get_input_embeddings()(ids[b: b + 1]).detach().requires_grad_()
            last = model(inputs_embeds=e, use_cache=False).logits[0, -1]
            last.max().backward()
            R.append((e * e.grad).float().sum(-1)[0])
        return torch.stack(R)
    Rd = dropin()
    dt_d = timed(dropin, 3)
    cos = torch.nn.functional.cosine_similarity(Rd.double().flatten(), out["R_tok"].double().flatten(), dim=0)
    print(f"B = {B}: fused driver {dt_e * 1e3:8.1f} ms/step = {B / dt_e:6.2f} explanations/s | drop-in path {dt_d * 1e3:8.1f} ms = {B / dt_d:6.2f} "
          f"explanations/s | cosine(R fused, R drop-in) {float(cos):.4f}", flush=True)
