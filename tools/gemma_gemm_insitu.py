#!/usr/bin/env python3
"""Dev tool (GPU box): per-shape GEMM times INSIDE the Gemma-3-4B text step (bench.py's config-4 engine, 12 layers), with and without
ops.TAIL_SPLIT: HIP-event spans of ops.KernelTimer grouped by (flops, tag)."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench  # noqa: E402
import lxt_amd.ops as ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    eng, g, V = bench.config4_text_engine(dev, torch.bfloat16, L=12)
    ids = torch.randint(0, V, (8, 2048), generator=torch.Generator().manual_seed(99)).to(dev)
    for flag in (False, True, False, True):
        ops.TAIL_SPLIT = flag
        eng.explain(ids[:4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.explain(ids[4:])
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 3
        timer = ops.KernelTimer()
        ops.GEMM_TIMER = timer
        eng.explain(ids[4:])
        torch.cuda.synchronize()
        ops.GEMM_TIMER = None
        grp = collections.OrderedDict()
        seq = []
        for fl, e0, e1, tag in timer.records:
            ms = e0.elapsed_time(e1)
            grp.setdefault((fl, tag), []).append(ms)
            seq.append((fl, tag, ms))
        print(f"== TAIL_SPLIT={flag}: step {el * 1e3:.2f} ms untimed; {len(timer.records)} spans, {sum(sum(v) for v in grp.values()):.2f} ms in GEMM spans")
        for (fl, tag), v in grp.items():
            print(f"   {fl / 1e9:9.1f} GFLOP {tag:10s} x{len(v):3d}: mean {sum(v) / len(v) * 1e3:7.1f} us  min {min(v) * 1e3:7.1f}  -> {fl / (sum(v) / len(v)) * 1e-9:6.0f} TF/s")
        if flag:
            print("   layer-1 order:", " | ".join(f"{fl / 1e9:.0f}:{tag}:{ms * 1e3:.0f}us" for fl, tag, ms in seq[:12]))
    ops.TAIL_SPLIT = True


if __name__ == "__main__":
    main()
