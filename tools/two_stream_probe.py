#!/usr/bin/env python3
"""Dev probe: does running the step as TWO half-batches on two HIP streams (row / attention kernels of one half under the GEMMs of the
other) beat one batch of 4 prompts on one stream?  Llama-3-8B shape, S = 2048, bf16, two engines sharing nothing but the weights' values."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench as BN  # noqa: E402
import lxt_amd.engine as E  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(BN.LLAMA3_8B)
L = int(os.environ.get("LAYERS", 32))
cfg["n_layers"] = L
S, steps = 2048, 6


def make():
    W = BN.synth_weights(cfg, dev, torch.bfloat16, seed=0)
    eng = E.LlamaLRP(cfg, W, dtype=torch.bfloat16, device=dev, mode="efficient", max_seq=S)
    del W
    torch.cuda.empty_cache()
    return eng


e4, ea, eb = make(), make(), make()
ids = torch.randint(0, cfg["vocab"], (4 * (steps + 2), S), generator=torch.Generator().manual_seed(1)).to(dev)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def one(i):
    return e4.explain(ids[4 * i: 4 * i + 4])["R_tok"]


def two(i, graph=False):
    cur = torch.cuda.current_stream(dev)
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        ra = ea.explain(ids[4 * i: 4 * i + 2], graph=graph)["R_tok"]
    with torch.cuda.stream(sb):
        rb = eb.explain(ids[4 * i + 2: 4 * i + 4], graph=graph)["R_tok"]
    cur.wait_stream(sa); cur.wait_stream(sb)
    return torch.cat([ra, rb])


def timed(f, **kw):
    for i in range(2):
        r = f(i, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2, 2 + steps):
        r = f(i, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, r


t1, r1 = timed(one)
t2, r2 = timed(two)
t3, r3 = timed(two, graph=True)
err = float((r1 - r2).abs().max() / r1.abs().max())
print(f"{L} layers: one stream, 4 prompts: {t1 * 1e3:.1f} ms/step = {4 / t1:.2f} expl/s | two streams x 2 prompts: {t2 * 1e3:.1f} ms = {4 / t2:.2f} expl/s"
      f" | two streams, hipGraph replay: {t3 * 1e3:.1f} ms = {4 / t3:.2f} expl/s | max diff of the last step's R {err:.1e}", flush=True)
