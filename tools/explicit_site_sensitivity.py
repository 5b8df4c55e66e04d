#!/usr/bin/env python3
"""Dev tool (GPU box): which eps site of the explicit composite carries the fp32 disagreement at the BASELINE width?
For the instance of tests/test_baseline_size_gpu.py, switches ON one stabiliser site at a time (all others 0 = efficient
arithmetic), in the engine and in the oracle alike, and prints engine-fp32 vs oracle-fp64 next to oracle-fp32 vs oracle-fp64."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import llama as ol  # noqa: E402
from tests.util import nmax  # noqa: E402
import tests.test_baseline_size_gpu as T  # noqa: E402
import lxt_amd.engine as E  # noqa: E402

SITES = ["lin", "add", "qk", "mask", "pv", "rope", "ALL"]


def main():
    """usage: explicit_site_sensitivity.py [WSEED IDSEED]   (default: the first instance of tests/test_baseline_size_gpu.py)"""
    ws, is_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else T.SEEDS[0]
    print(f"# one stabiliser site at a time, H4096/I14336/d128/S2048, 2 layers, seeds ({ws},{is_}); {torch.get_num_threads()} host threads", flush=True)
    cfg = dict(T.CFG)
    W = ol.random_weights(cfg, seed=ws)
    ids = torch.randint(0, cfg["vocab"], (T.S,), generator=torch.Generator().manual_seed(is_))
    caches = {}
    for dt in (torch.float64, torch.float32):
        Wd = ol.cast_weights(W, dt)
        emb = Wd["embed"][ids]
        caches[dt] = (Wd, emb, ol.forward(cfg, Wd, emb))
    idx = int(caches[torch.float64][2]["logits_last"].argmax())
    eng = E.LlamaLRP(cfg, W, dtype=torch.float32, mode="explicit", max_seq=T.S)
    full = dict(ol.EXPLICIT)
    for site in SITES:
        table = dict(full) if site == "ALL" else {k: (full[k] if k == site else 0.0) for k in full}
        ol.EXPLICIT.clear()
        ol.EXPLICIT.update(table)
        ref = {}
        for dt, (Wd, emb, cache) in caches.items():
            G, layer_R = ol.backward(cfg, Wd, cache, idx, "explicit")
            ref[dt] = (emb * G).sum(-1)
        eng.eps = dict(table)
        eng.eps_g = table["lin"]
        t0 = time.time()
        out = eng.explain(ids[None], target=torch.tensor([idx]))
        print(f"site {site:5s}: engine32 vs oracle64 {nmax(out['R_tok'][0], ref[torch.float64]):.2e} | oracle32 vs oracle64 "
              f"{nmax(ref[torch.float32], ref[torch.float64]):.2e} | engine32 vs oracle32 {nmax(out['R_tok'][0], ref[torch.float32]):.2e}"
              f"  ({time.time() - t0:.1f}s)", flush=True)
    ol.EXPLICIT.clear()
    ol.EXPLICIT.update(full)


if __name__ == "__main__":
    main()
