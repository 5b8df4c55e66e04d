#!/usr/bin/env python3
"""Dev tool (GPU box): per-site explicit sensitivity on the small ragged test instance (S=333, prompt 1)."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import llama as ol
from tests.util import nmax
import lxt_amd.engine as E
cfg = dict(hidden=256, inter=512, n_layers=2, n_heads=8, n_kv=2, head_dim=32, vocab=512, rope_theta=5e5, rms_eps=1e-5)
W = ol.random_weights(cfg, seed=302)
S = 333
ids = torch.randint(0, 512, (2, S), generator=torch.Generator().manual_seed(S))[1]
caches = {}
for dt in (torch.float64, torch.float32):
    Wd = ol.cast_weights(W, dt)
    emb = Wd["embed"][ids]
    caches[dt] = (Wd, emb, ol.forward(cfg, Wd, emb))
idx = int(caches[torch.float64][2]["logits_last"].argmax())
full = dict(ol.EXPLICIT)
for sparse in (True, False):
    eng = E.LlamaLRP(cfg, W, dtype=torch.float32, mode="explicit", max_seq=512, sparse_top=sparse)
    for site in ["lin", "add", "qk", "mask", "pv", "rope", "ALL"]:
        table = dict(full) if site == "ALL" else {k: (full[k] if k == site else 0.0) for k in full}
        ol.EXPLICIT.clear(); ol.EXPLICIT.update(table)
        ref = {}
        for dt, (Wd, emb, cache) in caches.items():
            G, _ = ol.backward(cfg, Wd, cache, idx, "explicit")
            ref[dt] = (emb * G).sum(-1)
        eng.eps = dict(table); eng.eps_g = table["lin"]
        out = eng.explain(ids[None], target=torch.tensor([idx]))
        d = (out["R_tok"][0].double().cpu() - ref[torch.float64]).abs() / ref[torch.float64].abs().max()
        print(f"sparse_top={sparse} site {site:5s}: engine32 vs oracle64 {float(d.max()):.2e} (token {int(d.argmax())}) | oracle32 vs oracle64 {nmax(ref[torch.float32], ref[torch.float64]):.2e}", flush=True)
ol.EXPLICIT.clear(); ol.EXPLICIT.update(full)
