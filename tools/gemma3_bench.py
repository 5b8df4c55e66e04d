#!/usr/bin/env python3
"""BASELINE config 4 timing: Gemma-3-4B-it SHAPE (random init, generated on the device), image + text, bf16, through
lxt_amd.efficient.monkey_patch (drop-in path, autograd-driven; sdpa -> the HIP attention also inside SigLIP, head_dim 72
zero-padded to 128).  One explanation = forward + LRP backward + relevance of the 4096 ViT patches and of the text tokens.
Dims from the public model card (not confirmable offline).  Dev tool; not the judged benchmark."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
warnings.simplefilter("ignore")
from transformers import Gemma3Config, Gemma3ForConditionalGeneration
from transformers.models.gemma3 import modeling_gemma3
from lxt_amd.efficient import monkey_patch

monkey_patch(modeling_gemma3)
L = int(os.environ.get("G3_TEXT_LAYERS", 34))
LV = int(os.environ.get("G3_VIT_LAYERS", 27))
text = dict(hidden_size=2560, intermediate_size=10240, num_hidden_layers=L, num_attention_heads=8, num_key_value_heads=4, head_dim=256,
            vocab_size=262208, sliding_window=1024, layer_types=[("full_attention" if (i + 1) % 6 == 0 else "sliding_attention") for i in range(L)],
            max_position_embeddings=8192, query_pre_attn_scalar=256)
vision = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=LV, num_attention_heads=16, image_size=896, patch_size=14,
              num_channels=3, vision_use_head=False)
cfg = Gemma3Config(text_config=text, vision_config=vision, mm_tokens_per_image=256, image_token_id=262144, boi_token_id=255999,
                   eoi_token_id=256000, attn_implementation="sdpa")
torch.manual_seed(0)
with torch.device("cuda"):
    model = Gemma3ForConditionalGeneration(cfg).to(torch.bfloat16).eval()
for p in model.parameters():
    p.requires_grad_(False)
model.model.multi_modal_projector.mm_input_projection_weight.normal_(0, 0.02)
print(f"params {sum(p.numel() for p in model.parameters())/1e9:.2f} B", flush=True)
for S_text in (256, 1792):
    S = S_text + 258
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 255000, (1, S), generator=g)
    ids[0, 4] = 255999
    ids[0, 5:261] = 262144
    ids[0, 261] = 256000
    ids = ids.cuda()
    tt = (ids == 262144).long()
    pv = torch.randn(1, 3, 896, 896, generator=g).cuda().bfloat16()

    def run():
        e = model.get_input_embeddings()(ids).detach().requires_grad_()
        px = pv.clone().requires_grad_()
        last = model(inputs_embeds=e, pixel_values=px, token_type_ids=tt, use_cache=False).logits[0, -1]
        last.max().backward()
        R_tok = (e * e.grad).float().sum(-1)
        R_patch = (px * px.grad).float().reshape(3, 64, 14, 64, 14).sum((0, 2, 4))      # 64 x 64 = 4096 ViT patches
        return R_tok, R_patch
    for _ in range(2):
        t1 = time.perf_counter()
        R_tok, R_patch = run()
        torch.cuda.synchronize()
        print(f"  warm-up run {time.perf_counter() - t1:.2f} s", flush=True)
    assert torch.isfinite(R_tok).all() and torch.isfinite(R_patch).all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"Gemma-3-4B-it shape, image (4096 patches -> 256 tokens) + {S_text} text tokens (S={S}), bf16, drop-in path: "
          f"{dt*1e3:8.1f} ms/explanation  {1/dt:6.2f} explanations/s   sum R text {float(R_tok.sum()):.4f} image {float(R_patch.sum()):.4f}", flush=True)
