#!/usr/bin/env python3
"""BASELINE config 2 timing: BERT-base (random init), S=128, fp32, full-model relevance through
lxt_amd.efficient.monkey_patch (drop-in path, autograd-driven).  Dev tool; not the judged benchmark."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
warnings.simplefilter("ignore")
from tests.golden.hf_models import build_bert
from transformers.models.bert import modeling_bert
from lxt_amd.efficient import monkey_patch

monkey_patch(modeling_bert)
for dtype in (torch.float32, torch.bfloat16):
    model = build_bert(seed=0, attn="sdpa").to(dtype).cuda()
    for p in model.parameters():
        p.requires_grad_(False)
    for B in (1, 16, 64):
        ids = torch.randint(0, 30522, (B, 128), generator=torch.Generator().manual_seed(1)).cuda()

        def run():
            e = model.get_input_embeddings()(ids).requires_grad_()
            logits = model(inputs_embeds=e).logits
            logits.max(-1).values.sum().backward()
            return (e * e.grad).float().sum(-1)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            R = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"BERT-base S=128 {str(dtype)[6:]:8s} batch {B:3d}: {dt*1e3:8.2f} ms/step  {B/dt:9.1f} explanations/s", flush=True)

# ---- the fused driver (lxt_amd.engine_bert.BertLRP): eager launches and hipGraph replay
from lxt_amd.engine_bert import BertLRP
for dtype in (torch.float32, torch.bfloat16):
    model = build_bert(seed=0, attn="eager")
    for mode in ("efficient", "explicit"):
        eng = BertLRP.from_hf(model, dtype=dtype, mode=mode)
        for B in (1, 16, 64):
            ids = torch.randint(0, 30522, (B, 128), generator=torch.Generator().manual_seed(1)).cuda()
            for graph in (False, True):
                for _ in range(3):
                    eng.explain(ids, graph=graph)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 20
                for _ in range(n):
                    eng.explain(ids, graph=graph)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                print(f"BertLRP {mode:9s} {str(dtype)[6:]:8s} batch {B:3d} {'hipGraph' if graph else 'eager   '}: {dt*1e3:8.2f} ms/step  "
                      f"{B/dt:9.1f} explanations/s", flush=True)
