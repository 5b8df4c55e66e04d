#!/usr/bin/env python3
"""Which part of the fused flow (K1n epilogues, folded attn_bwd_prep, RoPE backward in the dQ store, gated coefficient stash) moves the bf16 engine's
error against the fp32 engine?  VERDICT r5 weak item 1: fused 1.11e-2 vs stand-alone 6.4e-3 on test_llama_bf16_norm_folded_into_gemms' instance.
Every configuration on the SAME folded bf16 weights, against the fp32 engine on the unfolded weights: normalised max error and relative L2 error
of the token relevance, per prompt, over several id seeds (one instance is one draw)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lxt_amd  # noqa: E402,F401
import lxt_amd.engine as E  # noqa: E402
import lxt_amd.ops as ops  # noqa: E402
from oracle import llama as ol  # noqa: E402

cfg = dict(hidden=2048, inter=5632, n_layers=3, n_heads=16, n_kv=4, head_dim=128, vocab=1024, rope_theta=1e4, rms_eps=1e-5)
W = ol.random_weights(cfg, seed=77)
g = torch.Generator().manual_seed(78)
for L in W["layers"]:
    L["ln1"] = (0.25 + 1.5 * torch.rand(cfg["hidden"], generator=g))
    L["ln2"] = (0.25 + 1.5 * torch.rand(cfg["hidden"], generator=g))
B, S = 3, 2048
CONFIGS = [("all fused", dict()), ("no K1n (norm stand-alone)", dict(NORM_FUSION=False)),
           ("K1n fwd only", dict(NORM_FUSION=frozenset({"fwd"}))), ("K1n bwd_qkv only", dict(NORM_FUSION=frozenset({"bwd_qkv"}))),
           ("K1n bwd_gu only", dict(NORM_FUSION=frozenset({"bwd_gu"}))), ("no prep fusion", dict(PREP_FUSION=False)),
           ("no rope-bwd fusion", dict(ROPE_BWD_FUSION=False)), ("no gated fusion", dict(GATED_FUSION=False)),
           ("nothing fused", dict(NORM_FUSION=False, PREP_FUSION=False, ROPE_BWD_FUSION=False, GATED_FUSION=False))]
ref_eng = E.LlamaLRP(cfg, W, dtype=torch.float32, mode="efficient", max_seq=S)
res = {n: [] for n, _ in CONFIGS}
res["unfolded, nothing fused"] = []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    ids = torch.randint(0, cfg["vocab"], (B, S), generator=torch.Generator().manual_seed(100 + seed))
    ref = ref_eng.explain(ids)
    R0 = ref["R_tok"].double()
    ref_eng.release()

    def errs(out):
        R = out["R_tok"].double()
        return [(float((R[b] - R0[b]).abs().max() / R0[b].abs().max()), float((R[b] - R0[b]).norm() / R0[b].norm())) for b in range(B)]
    for fold in (True, False):
        eng = E.LlamaLRP(cfg, W, dtype=torch.bfloat16, mode="efficient", max_seq=S, sparse_top=False, fold_norm=fold)
        for name, kw in (CONFIGS if fold else [("unfolded, nothing fused", CONFIGS[-1][1])]):
            keep = {k: getattr(ops, k) for k in kw}
            try:
                for k, v in kw.items():
                    setattr(ops, k, v)
                eng._nf_cache.clear()
                res[name] += errs(eng.explain(ids, target=ref["idx"]))
            finally:
                for k, v in keep.items():
                    setattr(ops, k, v)
        eng.release()
        del eng
print(f"{'configuration':32s} {'nmax: gmean':>12s} {'max':>9s} | {'rel L2: gmean':>13s} {'max':>9s}   ({len(next(iter(res.values())))} prompts)")
for n, v in res.items():
    a = torch.tensor([x[0] for x in v]).double()
    l2 = torch.tensor([x[1] for x in v]).double()
    print(f"{n:32s} {float(a.log().mean().exp()):12.3e} {float(a.max()):9.2e} | {float(l2.log().mean().exp()):13.3e} {float(l2.max()):9.2e}")
