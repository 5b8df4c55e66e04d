"""CPU: the committed fp64-oracle cache the BERT GPU tests read (tests/golden/bert_oracle_cache.npz, tests/util.py: bert_oracle) is what
oracle/bert.py computes -- one cached case is recomputed live (S = 37, efficient and explicit placement)."""
import torch

from tests.golden import bert_explicit_compose as C
from tests.golden.hf_models import build_bert, wsum
from tests.util import bert_oracle, nmax
import tests.util as U


def test_cached_bert_oracle_matches_live_oracle():
    model = build_bert(seed=0, attn="eager")
    W64 = C.weights_from_hf(model, torch.float64)
    ids = torch.randint(0, model.config.vocab_size, (1, 37), generator=torch.Generator().manual_seed(37))[0]
    for eps_zero, draws in ((True, 0), (False, 2)):
        hit = None
        for target in (0, 1):
            c = bert_oracle(W64, ids, target, eps_zero=eps_zero, draws=draws, wsum_=wsum(model), compute=False)
            if c is not None:
                hit = (c, target)
        assert hit, "the S = 37 case is missing from the cache: re-run tests/golden/make_golden_bert_oracle_cache.py"
        cached, target = hit
        saved, U._BERT_CACHE = U._BERT_CACHE, {}
        try:
            live = bert_oracle(W64, ids, target, eps_zero=eps_zero, draws=draws)
        finally:
            U._BERT_CACHE = saved
        assert cached["cached"] and not live["cached"]
        assert nmax(cached["R_tok"], live["R_tok"]) < 1e-9 and abs(cached["logit"] - live["logit"]) < 1e-12
        assert nmax(cached["layer_R"], live["layer_R"]) < 1e-9
