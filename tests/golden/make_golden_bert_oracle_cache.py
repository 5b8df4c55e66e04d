#!/usr/bin/env python3
"""Pre-computes the fp64 oracle results (oracle/bert.py in fp64) the BERT GPU tests compare against, for the
exact cases those tests run -> tests/golden/bert_oracle_cache.npz (tests/util.py: bert_oracle).  CPU only; needs neither the GPU nor
the reference: the oracle itself is pinned against the reference by the bert_base*.npz fixtures.

    python tests/golden/make_golden_bert_oracle_cache.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from oracle import bert as ob  # noqa: E402
from tests.golden import bert_explicit_compose as C  # noqa: E402
from tests.golden.hf_models import build_bert, wsum  # noqa: E402
from tests.util import bert_oracle, _bert_key, load, t  # noqa: E402

model = build_bert(seed=0, attn="eager")
W64 = C.weights_from_hf(model, torch.float64)
out = {"wsum": np.float64(wsum(model))}


def add(ids, target, eps_zero, draws, rel=1e-7):
    if target is None:
        target = int(ob.forward(ob.cast(W64, torch.float64), ids, None)["logits"].argmax())
    key = _bert_key(ids, target, eps_zero, draws, rel)
    if key + "/R_tok" in out:
        return
    r = bert_oracle(W64, ids, target, eps_zero=eps_zero, draws=draws, rel=rel)
    out[key + "/R_tok"] = r["R_tok"].double().numpy()
    out[key + "/logit"] = np.float64(r["logit"])
    out[key + "/layer_R"] = np.asarray(r["layer_R"], dtype=np.float64)
    print(f"{key}: logit {r['logit']:+.6f}", flush=True)


# tests/test_bert_engine_gpu.py::test_bert_engine_explicit_fp32_vs_reference_and_oracle
fx = load("bert_base_explicit.npz")
add(t(fx["ids"]), int(fx["idx"]), False, 3)
# ::test_bert_engine_ragged_lengths_vs_oracle
for B, S in [(1, 37), (3, 100), (2, 192)]:
    ids = torch.randint(0, model.config.vocab_size, (B, S), generator=torch.Generator().manual_seed(S))
    for b in range(B):
        add(ids[b], None, True, 0)
    add(ids[0], None, False, 2)
# tests/hf_family_worker.py::bert_explicit_padded
S, lens = 128, (128, 100)
ids = torch.randint(0, model.config.vocab_size, (2, S), generator=torch.Generator().manual_seed(11))
for b, L in enumerate(lens):
    add(ids[b, :L], None, False, 3)
np.savez_compressed(os.path.join(HERE, "bert_oracle_cache.npz"), **out)
print("wrote", len(out), "arrays")
