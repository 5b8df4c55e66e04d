#!/usr/bin/env python3
"""Fixture generator (build container only: imports the REAL reference from /root/reference).

BASELINE config 2 in EXPLICIT semantics: BERT-base (random init, seed 0), S = 128, the reference's explicit composite
hand-composed from lxt.explicit.functional / lxt.explicit.rules (tests/golden/bert_explicit_compose.py -- the vendored
lxt/explicit/models/bert.py cannot be imported under transformers 5.x).  Freezes idx, logit, per-token and per-neuron relevance
in fp32 and fp64 plus the reference's own fp32-vs-fp64 gap, and asserts the repo's oracle (oracle/bert.py) against it."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import lxt.explicit.functional as lf  # noqa: E402
import lxt.explicit.rules as rules  # noqa: E402

from oracle import bert as ob  # noqa: E402
from tests.golden import bert_explicit_compose as C  # noqa: E402
from tests.golden.hf_models import build_bert, wsum  # noqa: E402
from tests.util import nmax  # noqa: E402


def main():
    model = build_bert(seed=0, attn="eager")
    ids = torch.randint(0, model.config.vocab_size, (1, 128), generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        hf_logits = model(input_ids=ids).logits[0]
    out = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        W = C.weights_from_hf(model, dt)
        r = C.explain(lf, rules, W, ids)
        assert nmax(r["logits"], hf_logits) < 1e-5, "composite forward != HF forward"
        out[name] = r
        o = ob.explain(W, ids[0], target=r["idx"], dtype=dt)
        print(f"[{name}] idx {r['idx']} logit {r['logit']:.6f} sum R {float(r['R_tok'].sum()):.6f} | oracle vs reference: token "
              f"{nmax(o['R_tok'], r['R_tok']):.2e} neuron {nmax(o['R_emb'], r['R_emb']):.2e}")
        assert nmax(o["R_tok"], r["R_tok"]) < (1e-9 if dt == torch.float64 else 2e-3)
    gap = nmax(out["f32"]["R_tok"], out["f64"]["R_tok"])
    print(f"reference's own fp32-vs-fp64 gap: token {gap:.2e} neuron {nmax(out['f32']['R_emb'], out['f64']['R_emb']):.2e}")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bert_base_explicit.npz"), ids=ids[0].numpy(), wsum=wsum(model),
                        idx=out["f64"]["idx"], logit=out["f64"]["logit"], logits=out["f64"]["logits"].numpy(),
                        R_tok=out["f32"]["R_tok"].numpy(), R_tok_fp64=out["f64"]["R_tok"].numpy(),
                        R_emb_fp64=out["f64"]["R_emb"].numpy().astype(np.float32), cond_gap=gap)


if __name__ == "__main__":
    main()
