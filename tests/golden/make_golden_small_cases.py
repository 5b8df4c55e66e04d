#!/usr/bin/env python3
"""Reference-run yardsticks for the SMALL explicit-mode GPU cases (VERDICT r4 "what's weak" 2: their bars came from a builder-made noise
model, tests/util.fp32_conditioning*).  Run in the build container only (needs /root/reference):

    PYTHONPATH=/root/reference:/root/repo python tests/golden/make_golden_small_cases.py

Every explicit-mode case of tests/test_engine_gpu.py (ragged lengths, left-padded batch, dense / contrastive seed), of
tests/test_bert_engine_gpu.py (ragged lengths, first prompt) and of tests/hf_family_worker.py (right-padded explicit BERT batch) is run
through the REFERENCE's own Functions -- Llama: lxt.explicit.functional / rules / modules composed as lxt/explicit/models/llama.py:83-93,
226-260,273-281,379-391,481-488 (make_golden.ref_explicit_llama); BERT: the composition of lxt/explicit/models/bert.py:60-65,249-253,
338-373,396 (bert_explicit_compose over lxt.explicit.functional / rules) -- in fp32 and in fp64 on the un-padded prompt, and compared with
the exact result (the repo's oracle in pure fp64, which the script also pins against the reference's fp64 run).  Stored per case key:

    <key>/idx, <key>/logit      explained class / token and its logit (fp64)
    <key>/R_tok                 exact per-token relevance (fp64)
    <key>/gap                   the REFERENCE's fp32 run against it (normalised max error): the tests' yardstick (bar = max(1e-4, 3 gap))
    <key>/gap64                 the reference's "fp64" run against it (Llama: lf.rms_norm_identity evaluates the norm in fp32 whatever the
                                dtype, lxt/explicit/functional.py:481-486, so this is not 0)
Outputs only; no reference source is stored."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import lxt.explicit.functional as lf            # noqa: E402  (the reference)
import lxt.explicit.rules as lrules             # noqa: E402
from make_golden import ref_explicit_llama, nmax  # noqa: E402
from oracle import llama as ol                  # noqa: E402
from oracle import bert as ob                   # noqa: E402
from tests.golden import bert_explicit_compose as C   # noqa: E402
from tests.golden.hf_models import build_bert, wsum    # noqa: E402

OUT = {}


def llama_case(key, cfg, W, ids, seed_fn=None):
    """one un-padded prompt; seed_fn(logits_last fp64, idx) -> relevance pattern [V] for .backward (None: the explained logit)"""
    W64 = ol.cast_weights(W, torch.float64)
    ex = ol.explain(cfg, W, ids=ids, mode="explicit", dtype=torch.float64)
    idx = ex["idx"]
    rel = None
    if seed_fn is not None:       # explicit mode seeds a RELEVANCE pattern over the logits: mask * logits (oracle.llama.backward, same convention)
        rel = seed_fn(ex["logits_last"].double(), idx) * ex["logits_last"].double()
        ex = ol.explain(cfg, W, ids=ids, mode="explicit", dtype=torch.float64, seed=rel)
    r64 = ref_explicit_llama(cfg, W64, W64["embed"][ids], target=idx, seed=rel)
    r32 = ref_explicit_llama(cfg, W, W["embed"][ids], target=idx, seed=None if rel is None else rel.float())
    gap, gap64 = nmax(r32["R_tok"], ex["R_tok"]), nmax(r64["R_tok"], ex["R_tok"])
    assert gap64 < 1e-3, (key, gap64, gap)       # the reference in "fp64" (fp32 norms) vs the pure-fp64 oracle: the same composition
    OUT[f"{key}/idx"], OUT[f"{key}/logit"] = idx, float(ex["logit"])
    OUT[f"{key}/R_tok"], OUT[f"{key}/gap"], OUT[f"{key}/gap64"] = ex["R_tok"].double().numpy(), gap, gap64
    print(f"{key}: idx {idx}  reference fp32 vs exact {gap:.2e}  reference 'fp64' vs exact {gap64:.2e}", flush=True)


def llama_cases():
    cfg2 = dict(hidden=256, inter=512, n_layers=2, n_heads=8, n_kv=2, head_dim=32, vocab=512, rope_theta=5e5, rms_eps=1e-5)
    cfg3 = dict(cfg2, n_layers=3)
    # tests/test_engine_gpu.py::test_llama_ragged_lengths
    W = ol.random_weights(cfg2, seed=302)
    for S, B in ((37, 1), (100, 3), (333, 2)):
        ids = torch.randint(0, 512, (B, S), generator=torch.Generator().manual_seed(S))
        for b in range(B):
            llama_case(f"llama_ragged_S{S}_b{b}", cfg2, W, ids[b])
    # ::test_llama_left_padded_batch (every prompt alone, un-padded)
    W = ol.random_weights(cfg3, seed=303)
    S, lens = 150, [150, 97, 31, 1]
    ids = torch.randint(0, 512, (len(lens), S), generator=torch.Generator().manual_seed(5))
    for b, n in enumerate(lens):
        llama_case(f"llama_leftpad_b{b}", cfg3, W, ids[b, S - n:])
    # ::test_llama_dense_seed_contrastive: +1 on the arg-max logit, -1/V elsewhere
    W = ol.random_weights(cfg3, seed=311)
    ids = torch.randint(0, 512, (2, 96), generator=torch.Generator().manual_seed(9))
    V = cfg3["vocab"]

    def contrast(logits, idx):
        m = torch.full((V,), -1.0 / V, dtype=torch.float64)
        m[idx] = 1.0
        return m
    for b in range(2):
        llama_case(f"llama_dense_seed_b{b}", cfg3, W, ids[b], seed_fn=contrast)


def bert_case(key, W32, W64, ids):
    r64 = C.explain(lf, lrules, W64, ids[None])
    r32 = C.explain(lf, lrules, W32, ids[None], target=r64["idx"])
    o64 = ob.explain(W64, ids, target=r64["idx"], dtype=torch.float64)
    gap, gap64 = nmax(r32["R_tok"], o64["R_tok"]), nmax(r64["R_tok"], o64["R_tok"])
    assert gap64 < 1e-9, (key, gap64)
    OUT[f"{key}/idx"], OUT[f"{key}/logit"] = r64["idx"], r64["logit"]
    OUT[f"{key}/R_tok"], OUT[f"{key}/gap"], OUT[f"{key}/gap64"] = o64["R_tok"].double().numpy(), gap, gap64
    print(f"{key}: idx {r64['idx']} logit {r64['logit']:+.6f}  reference fp32 vs exact {gap:.2e}  reference fp64 vs oracle fp64 {gap64:.1e}", flush=True)


def bert_cases():
    model = build_bert(seed=0, attn="eager")
    V = model.config.vocab_size
    W32, W64 = C.weights_from_hf(model, torch.float32), C.weights_from_hf(model, torch.float64)
    OUT["bert_wsum"] = wsum(model)
    # tests/test_bert_engine_gpu.py::test_bert_engine_ragged_lengths_vs_oracle (explicit: first prompt)
    for B, S in ((1, 37), (3, 100), (2, 192)):
        ids = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(S))
        bert_case(f"bert_ragged_S{S}_b0", W32, W64, ids[0])
    # tests/hf_family_worker.py::bert_explicit_padded (right-padded batch; every row alone, un-padded)
    S, lens = 128, (128, 100)
    ids = torch.randint(0, V, (2, S), generator=torch.Generator().manual_seed(11))
    for b, L in enumerate(lens):
        bert_case(f"bert_padded_b{b}", W32, W64, ids[b, :L])


if __name__ == "__main__":
    torch.set_num_threads(8)
    llama_cases()
    bert_cases()
    np.savez_compressed(os.path.join(HERE, "small_cases_ref.npz"), **{k: np.asarray(v) for k, v in OUT.items()})
    print(f"wrote small_cases_ref.npz ({len(OUT)} arrays)")
