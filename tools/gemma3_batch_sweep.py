#!/usr/bin/env python3
"""Dev tool (GPU box): Gemma-3-4B text step (bench.py's config-4 engine) at 2 / 4 / 8 prompts per step: explanations/s, host time to issue a
step against its wall time -- is the driver (no arena, no hipGraph) launch-bound?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    eng, g, V = bench.config4_text_engine(dev, torch.bfloat16)
    for B in (2, 4, 8):
        ids = torch.randint(0, V, (2 * B, 2048), generator=torch.Generator().manual_seed(99)).to(dev)
        eng.explain(ids[:B])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for i in range(n):
            eng.explain(ids[B:])
        ti = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"B = {B}: {el / n * 1e3:7.1f} ms per step = {B * n / el:6.2f} explanations/s; host issue {ti / n * 1e3:6.1f} ms per step", flush=True)


if __name__ == "__main__":
    main()
