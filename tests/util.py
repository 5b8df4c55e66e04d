"""Shared helpers for the tests (metric of SURVEY.md 8d)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def nmax(a, b):
    """normalised max error  max|a-b| / max|b|  (the parity metric; element-wise relative
    error is unusable near zero relevance -- SURVEY.md finding 3)."""
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def t(a, dtype=None):
    x = torch.from_numpy(np.asarray(a))
    return x.to(dtype) if dtype is not None else x


def llama_case(name):
    """-> (cfg, W, ids, fixture) ; weights regenerated from the recorded seed (or carried)."""
    from oracle import llama as ol
    fx = load(f"llama_{name}.npz")
    cfg = {k: (float(v) if k in ("rope_theta", "rms_eps") else int(v))
           for k, v in zip(fx["cfg_keys"].tolist(), fx["cfg_vals"].tolist())}
    W = ol.random_weights(cfg, seed=int(fx["wseed"]))
    tot = float(W["embed"].double().abs().sum() + W["lm_head"].double().abs().sum())
    for L in W["layers"]:
        tot += sum(float(v.double().abs().sum()) for v in L.values())
    assert abs(tot - float(fx["wsum"])) <= 1e-9 * abs(tot), "synthetic weights did not reproduce"
    if "W_embed" in fx:  # self-contained fixture: the carried weights must equal the regenerated ones
        assert np.array_equal(fx["W_embed"], W["embed"].numpy())
    return cfg, W, t(fx["ids"]), fx


def fp32_conditioning(cfg, W, ids, target, mode, ref64=None, draws=3, rel=3e-7, seed=None, emb=None):
    """How far can an fp32 evaluation of THIS instance be from the exact result?  The explicit rules multiply by z/(z + eps),
    which has a pole at z = -eps: an activation that lands within a fraction of a percent of it turns an fp32-rounding-sized
    perturbation of z into an O(1) change of that element's relevance, and whether a given fp32 implementation (the reference's
    own included) is hit depends on its summation order -- one implementation's fp32-vs-fp64 gap is therefore NOT a stable scale
    for another's.  This estimates the scale itself: the fp64 oracle is re-run with every stored activation multiplied by
    (1 + rel * N(0,1)) (rel ~ a few fp32 ulps: storage rounding + a K-term accumulation) and the largest normalised deviation of
    the token relevance over `draws` noise seeds is returned.  ~1e-7 on well-conditioned instances (and always for mode
    'efficient'), 1e-4 ... 1e-2 where a pole is near."""
    from oracle import llama as ol
    if ref64 is None:
        ref64 = ol.explain(cfg, W, ids=ids, emb=emb, target=target, mode=mode, dtype=torch.float64, seed=seed)["R_tok"]
    worst = 0.0
    for d in range(draws):
        g = torch.Generator().manual_seed(1000 + d)
        noisy = ol.explain(cfg, W, ids=ids, emb=emb, target=target, mode=mode, dtype=torch.float64, seed=seed,
                           rnd=lambda x: x * (1 + rel * torch.randn(x.shape, generator=g, dtype=x.dtype)))
        worst = max(worst, nmax(noisy["R_tok"], ref64))
    return worst


def fp32_conditioning_bert(W64, ids, target, ref64=None, draws=3, rel=3e-7):
    """fp32_conditioning for the explicit BERT composite (oracle/bert.py).  The noise of a stored activation is relative to its
    ROW's scale, not to the element: LayerNorm outputs and residual sums reach |y| ~ 1e-6 (the LayerNormEpsilon stabiliser!) by
    cancellation of O(1) terms, so their fp32 error is ~1e-7 absolute however small they are."""
    from oracle import bert as ob
    if ref64 is None:
        ref64 = ob.explain(W64, ids, target=target, dtype=torch.float64)["R_tok"]
    worst = 0.0
    for d in range(draws):
        g = torch.Generator().manual_seed(2000 + d)

        def rnd(x):
            scale = x.pow(2).mean(-1, keepdim=True).sqrt()
            return x + rel * scale * torch.randn(x.shape, generator=g, dtype=x.dtype)
        worst = max(worst, nmax(ob.explain(W64, ids, target=target, dtype=torch.float64, rnd=rnd)["R_tok"], ref64))
    return worst
