#!/usr/bin/env python3
"""Llama-3-8B SHAPE through the DROP-IN path (lxt_amd.efficient.monkey_patch on a HuggingFace LlamaForCausalLM, autograd-
driven, the reference's quickstart protocol) next to the fused engine built from the same weights.  Dev tool; the judged
benchmark is bench.py (fused engine)."""
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
    model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
for p in model.parameters():
    p.requires_grad_(False)
S = 2048
eng = E.LlamaLRP.from_hf(model, mode="efficient", max_seq=S)
TARGET = {}                       # the fused engine picks the explained token (arg-max of ITS logits); the drop-in path explains the
for B in (1, 4):                  # same one -- with random-init weights the logits are nearly flat and bf16 noise moves the arg-max
    ids = torch.randint(0, 128256, (B, S), generator=torch.Generator().manual_seed(1234)).cuda()
    tgt = eng.explain(ids)["idx"].long()

    def run():
        e = model.get_input_embeddings()(ids).detach().requires_grad_()
        logits = model(inputs_embeds=e, use_cache=False, logits_to_keep=1).logits          # last position only, like the engine
        logits[torch.arange(B, device="cuda"), -1, tgt].sum().backward()
        return (e * e.grad).float().sum(-1)
    for _ in range(2):
        R = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 4
    for _ in range(n):
        R = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"drop-in path (HF LlamaForCausalLM + monkey_patch, sdpa -> HIP attention), {L} layers, bf16, S={S}, batch {B}: "
          f"{dt*1e3:8.1f} ms/step  {B/dt:6.2f} explanations/s", flush=True)
out = eng.explain(ids)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    out = eng.explain(ids)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 4
err = float((out["R_tok"] - R).abs().max() / R.abs().max())
print(f"fused engine from the same weights, batch 4: {dt*1e3:8.1f} ms/step  {4/dt:6.2f} explanations/s ; drop-in vs engine relevance (bf16) {err:.2e}")

# ---- precision: the same weights through the fp32 engine (fp32 GEMMs, the <= 1e-4 parity path of the tests) as the yardstick
# for the two bf16 paths at the full 32-layer depth (random init: worst-case conditioning, SURVEY finding 3)
if os.environ.get("FP32_CHECK", "1") == "1":
    R_drop, R_eng = R.float(), out["R_tok"].float()
    del eng
    torch.cuda.empty_cache()
    eng32 = E.LlamaLRP.from_hf(model, mode="efficient", max_seq=S, dtype=torch.float32)
    R32 = eng32.explain(ids, target=out["idx"])["R_tok"].float()

    def stats(a, b):
        nm = float((a - b).abs().max() / b.abs().max())
        cos = float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0))
        k = 20
        top = sum(len(set(a[i].abs().topk(k).indices.tolist()) & set(b[i].abs().topk(k).indices.tolist())) for i in range(a.shape[0])) / (k * a.shape[0])
        return f"normalised max err {nm:.2e}, cosine {cos:.5f}, top-{k} token overlap {top:.2f}"
    print("bf16 engine  vs fp32 engine:", stats(R_eng, R32))
    print("bf16 drop-in vs fp32 engine:", stats(R_drop, R32))
    print("bf16 engine  vs bf16 drop-in:", stats(R_eng, R_drop))
    # where does the HF-bf16 deviation come from?  variant: rotary embedding evaluated in fp32 (tables and arithmetic), result cast
    _orig = modeling_llama.apply_rotary_pos_emb

    def rope_fp32(q, k, cos, sin, *a, **kw):
        qe, ke = _orig(q.float(), k.float(), cos.float(), sin.float(), *a, **kw)
        return qe.to(q.dtype), ke.to(k.dtype)
    modeling_llama.apply_rotary_pos_emb = rope_fp32
    _rot = model.model.rotary_emb.forward

    def rot_fp32(x, position_ids, *a, **kw):
        c, s_ = _rot(x.float(), position_ids, *a, **kw)
        return c.float(), s_.float()
    model.model.rotary_emb.forward = rot_fp32
    print("bf16 drop-in with fp32 RoPE vs fp32 engine:", stats(run().float(), R32))
