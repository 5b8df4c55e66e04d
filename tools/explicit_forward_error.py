#!/usr/bin/env python3
"""Dev tool (GPU box): is the fp32 engine's FORWARD less accurate than a CPU fp32 evaluation (oracle in fp32)?  Compares the
stashed activations of the fused engine (fp32) with the oracle's fp64 cache, next to oracle-fp32 vs oracle-fp64, at the
BASELINE layer width; then the add / pv eps sites over three seeds (engine vs fp64 oracle, oracle fp32 vs fp64 oracle)."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import llama as ol  # noqa: E402
from tests.util import nmax  # noqa: E402
import tests.test_baseline_size_gpu as T  # noqa: E402
import lxt_amd.engine as E  # noqa: E402


def rel_err_stats(a, b):
    """max and rms of (a-b), both relative to rms(b) -- element errors in units of the tensor's scale"""
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    s = b.pow(2).mean().sqrt()
    d = (a - b)
    return float(d.abs().max() / s), float(d.pow(2).mean().sqrt() / s)


def main():
    cfg = dict(T.CFG)
    S = T.S
    full = dict(ol.EXPLICIT)
    for ws, is_ in ((20, 21), (22, 23), (24, 25)):
        W = ol.random_weights(cfg, seed=ws)
        ids = torch.randint(0, cfg["vocab"], (S,), generator=torch.Generator().manual_seed(is_))
        caches = {}
        for dt in (torch.float64, torch.float32):
            Wd = ol.cast_weights(W, dt)
            emb = Wd["embed"][ids]
            caches[dt] = (Wd, emb, ol.forward(cfg, Wd, emb))
        idx = int(caches[torch.float64][2]["logits_last"].argmax())
        eng = E.LlamaLRP(cfg, W, dtype=torch.float32, mode="explicit", max_seq=S, sparse_top=False)
        if ws == 20:
            emb32 = eng.embed.index_select(0, ids.to(eng.device))
            fw = eng.forward(emb32, 1, S)
            c64, c32 = caches[torch.float64][2]["layers"], caches[torch.float32][2]["layers"]
            H, d, nq, nk = cfg["hidden"], cfg["head_dim"], cfg["n_heads"], cfg["n_kv"]
            for li in range(2):
                st = fw["stash"][li]
                pairs = dict(h=(st["h"], c64[li]["h"], c32[li]["h"]),
                             a=(st["a"], c64[li]["a"], c32[li]["a"]), h1=(st["h1"], c64[li]["h1"], c32[li]["h1"]),
                             dn=(st["dn"], c64[li]["dn"], c32[li]["dn"]),
                             o=(st["o"], c64[li]["of"], c32[li]["of"]),
                             gate=(st["gu"][:, :cfg["inter"]], c64[li]["g"], c32[li]["g"]))
                for name, (e, r64, r32) in pairs.items():
                    me, re_ = rel_err_stats(e, r64)
                    mo, ro = rel_err_stats(r32, r64)
                    print(f"seed {ws} layer {li} {name:5s}: engine32 vs oracle64 max {me:.2e} rms {re_:.2e} | oracle32 vs oracle64 max {mo:.2e} rms {ro:.2e}", flush=True)
        for site in ("add", "pv"):
            table = {k: (full[k] if k == site else 0.0) for k in full}
            ol.EXPLICIT.clear()
            ol.EXPLICIT.update(table)
            ref = {}
            for dt, (Wd, emb, cache) in caches.items():
                G, _ = ol.backward(cfg, Wd, cache, idx, "explicit")
                ref[dt] = (emb * G).sum(-1)
            eng.eps = dict(table)
            eng.eps_g = table["lin"]
            out = eng.explain(ids[None], target=torch.tensor([idx]))
            d = (out["R_tok"][0].double().cpu() - ref[torch.float64]).abs() / ref[torch.float64].abs().max()
            top = torch.topk(d, 3)
            print(f"seed {ws} site {site:4s}: engine32 vs oracle64 {nmax(out['R_tok'][0], ref[torch.float64]):.2e} | oracle32 vs oracle64 "
                  f"{nmax(ref[torch.float32], ref[torch.float64]):.2e} | worst tokens {top.indices.tolist()} errs {[f'{x:.1e}' for x in top.values.tolist()]} "
                  f"median token err {float(d.median()):.1e}", flush=True)
        ol.EXPLICIT.clear()
        ol.EXPLICIT.update(full)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
