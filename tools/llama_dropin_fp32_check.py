#!/usr/bin/env python3
"""Full-size (Llama-3-8B shape, 32 layers, S=2048) parity of the two product paths in fp32: HF model + lxt_amd.efficient.monkey_patch
(autograd-driven drop-in) vs the fused engine built from the same weights.  Dev check."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
warnings.simplefilter("ignore")
from transformers import LlamaConfig, LlamaForCausalLM
from transformers.models.llama import modeling_llama
from lxt_amd.efficient import monkey_patch
import lxt_amd.engine as E

monkey_patch(modeling_llama)
L = int(os.environ.get("LAYERS", 32))
cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=L, num_attention_heads=32, num_key_value_heads=8,
                  vocab_size=128256, rms_norm_eps=1e-5, max_position_embeddings=8192, tie_word_embeddings=False,
                  rope_parameters=dict(rope_type="default", rope_theta=500000.0), attn_implementation="sdpa")
torch.manual_seed(0)
with torch.device("cuda"):
    model = LlamaForCausalLM(cfg).eval()           # fp32
for p in model.parameters():
    p.requires_grad_(False)
S, B = 2048, 2
ids = torch.randint(0, 128256, (B, S), generator=torch.Generator().manual_seed(1234)).cuda()
eng = E.LlamaLRP.from_hf(model, mode="efficient", max_seq=S)
out = eng.explain(ids)
tgt = out["idx"].long()
e = model.get_input_embeddings()(ids).detach().requires_grad_()
logits = model(inputs_embeds=e, use_cache=False, logits_to_keep=1).logits
print("logit engine vs drop-in:", out["logit"].tolist(), logits[torch.arange(B, device="cuda"), -1, tgt].tolist())
logits[torch.arange(B, device="cuda"), -1, tgt].sum().backward()
R = (e * e.grad).sum(-1)
nm = float((R - out["R_tok"]).abs().max() / out["R_tok"].abs().max())
cos = float(torch.nn.functional.cosine_similarity(R.flatten(), out["R_tok"].flatten(), dim=0))
print(f"fp32, {L} layers, S={S}: drop-in vs fused engine: normalised max err {nm:.2e}, cosine {cos:.7f}")
