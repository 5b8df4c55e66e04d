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


# ---- cached fp64 BERT oracle ----------------------------------------------------------------------------------------------------------
# The explicit BERT tests compare against oracle/bert.py in fp64 plus 2-3 noise draws of it (fp32_conditioning_bert): ~10 BERT-base
# passes in fp64 on the host per test, 30 s on an idle 128-thread host and 6 minutes on a busy one (measured: the round-3 suite went
# from 229 s to 1257 s on one box).  The results only depend on (seeded weights, ids, target, stabilisers), so they are computed once
# here in the build container by tests/golden/make_golden_bert_oracle_cache.py (the oracle alone, no reference needed) and committed;
# a miss (changed case) falls back to the live computation.
_BERT_CACHE = None


def _bert_key(ids, target, eps_zero, draws, rel):
    import hashlib
    h = hashlib.sha1(np.ascontiguousarray(ids.cpu().numpy().astype(np.int64)).tobytes()).hexdigest()[:16]
    return f"{h}_{int(target)}_{int(bool(eps_zero))}_{int(draws)}_{rel:g}"


def bert_oracle(W64, ids, target, eps_zero=False, draws=0, rel=1e-7, wsum_=None, compute=True):
    """-> dict(R_tok [S] f64, logit, layer_R, cond (0.0 when draws == 0)) of oracle/bert.py in fp64; eps_zero: every stabiliser 0 (the
    efficient placement).  Served from tests/golden/bert_oracle_cache.npz when the case is there."""
    global _BERT_CACHE
    import os
    from oracle import bert as ob
    key = _bert_key(ids, target, eps_zero, draws, rel)
    if _BERT_CACHE is None:
        path = os.path.join(os.path.dirname(__file__), "golden", "bert_oracle_cache.npz")
        _BERT_CACHE = dict(np.load(path)) if os.path.exists(path) else {}
    c = _BERT_CACHE
    if key + "/R_tok" in c and (wsum_ is None or abs(float(c["wsum"]) - wsum_) < 1e-6 * wsum_):
        return dict(R_tok=torch.as_tensor(c[key + "/R_tok"]), logit=float(c[key + "/logit"]), layer_R=torch.as_tensor(c[key + "/layer_R"]),
                    cond=float(c[key + "/cond"]), cached=True)
    if not compute:
        return None
    saved = dict(ob.EPS)
    try:
        if eps_zero:
            for k in ob.EPS:
                ob.EPS[k] = 0.0
        o64 = ob.explain(W64, ids, target=int(target), dtype=torch.float64)
        cond = fp32_conditioning_bert(W64, ids, int(target), o64["R_tok"], draws=draws, rel=rel) if draws else 0.0
    finally:
        ob.EPS.update(saved)
    return dict(R_tok=o64["R_tok"], logit=o64["logit"], layer_R=torch.as_tensor(o64["layer_R"]), cond=cond, cached=False)
