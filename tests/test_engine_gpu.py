"""GPU parity of the whole-model engine (lxt_amd.engine.LlamaLRP) against
  (1) the oracle run live on the host CPU (fp32 and fp64), and
  (2) the golden fixtures captured from the real reference (tests/golden/llama_*.npz):
      lxt.explicit (hand-composed from the reference's Functions) and lxt.efficient.
Metric: normalised max error max|dR|/max|R| over per-token (and per-neuron) relevance
(SURVEY.md 8d).  Bars: fp32 engine <= 1e-4 (north-star tolerance); bf16 engine <= 5e-2."""
import pytest
import torch

from oracle import llama as ol
from tests.util import nmax, llama_case, ref_case, ref_bar

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import lxt_amd.engine as e
    return e


@pytest.mark.parametrize("name", ["tiny", "mid", "d128"])
@pytest.mark.parametrize("mode", ["explicit", "efficient"])
def test_llama_fp32_vs_reference_and_oracle(eng_mod, name, mode):
    cfg, W, ids, fx = llama_case(name)
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode=mode, max_seq=512)
    out = eng.explain(ids[None], layer_relevance=True, return_G=True)
    assert int(out["idx"][0]) == int(fx["idx"])
    ref_key = "exp64_R_tok" if mode == "explicit" else "eff_R_tok"
    assert abs(float(out["logit"][0]) - float(fx["logit"])) < 1e-4
    e_tok = nmax(out["R_tok"][0], fx[ref_key])
    o64 = ol.explain(cfg, W, ids=ids, target=int(fx["idx"]), mode=mode, dtype=torch.float64)
    e_or = nmax(out["R_tok"][0], o64["R_tok"])
    e_neu = nmax((out["emb"][0].double() * out["G_emb"][0].double()), o64["R_emb"])
    e_lay = nmax(out["layer_R"][:, 0], o64["layer_R"])
    print(f"[{name}/{mode}] tok vs reference {e_tok:.2e} | tok vs oracle64 {e_or:.2e} | neuron {e_neu:.2e} | layer {e_lay:.2e}"
          f" | reference's own fp32-fp64 gap {float(fx['cond_gap']):.1e}")
    assert e_tok < 1e-4 and e_or < 1e-4 and e_neu < 1e-4 and e_lay < 1e-4


def test_llama_batch_equals_single(eng_mod):
    """prompts of a batch are independent: batched result == per-prompt result, bit for bit"""
    cfg, W, ids, fx = llama_case("mid")
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode="explicit", max_seq=512)
    g = torch.Generator().manual_seed(5)
    ids2 = torch.stack([ids, torch.randint(0, cfg["vocab"], ids.shape, generator=g)])
    both = eng.explain(ids2)
    for b in range(2):
        one = eng.explain(ids2[b:b + 1])
        assert torch.equal(one["R_tok"][0], both["R_tok"][b]) and int(one["idx"][0]) == int(both["idx"][b])


def test_llama_bf16_norm_folded_into_gemms(eng_mod):
    """K1n (round 5): the bf16 engine folds the per-layer RMSNorm weights into the consuming Linears (same network, same relevance under every
    rule) and, for M = B S rows in the efficient placement, runs the norms and residual sums inside the GEMM epilogues
    (ref lxt/efficient/patches.py:111-123; HF modeling_llama's h + attn, h + mlp).  A shape wide enough for the fused epilogues
    (>= 190 tiles per GEMM), NON-TRIVIAL norm weights: (1) fused and stand-alone flows agree to bf16 rounding, (2) both against the fp32
    engine on the UNFOLDED weights within the bf16 bar, (3) the explicit placement runs on the folded weights (stand-alone kernels, norm
    weight = 1) and agrees with the fp32 explicit engine the same way, (4) fused flow == itself under a hipGraph, batched == single."""
    import lxt_amd.ops as ops
    cfg = dict(hidden=2048, inter=5632, n_layers=3, n_heads=16, n_kv=4, head_dim=128, vocab=1024, rope_theta=1e4, rms_eps=1e-5)
    W = ol.random_weights(cfg, seed=77)
    g = torch.Generator().manual_seed(78)
    for L in W["layers"]:
        L["ln1"] = (0.25 + 1.5 * torch.rand(cfg["hidden"], generator=g))
        L["ln2"] = (0.25 + 1.5 * torch.rand(cfg["hidden"], generator=g))
    B, S = 3, 2048
    ids = torch.randint(0, cfg["vocab"], (B, S), generator=g)
    ref = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode="efficient", max_seq=S).explain(ids)
    tgt = ref["idx"]
    for sparse_top in (True, False):
        eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.bfloat16, mode="efficient", max_seq=S, sparse_top=sparse_top)
        assert eng.folded and eng._norm_fused(B * S) and not eng._norm_fused(S // 2)
        assert all(bool((Lw["ln1"] == 1).all()) and bool((Lw["ln2"] == 1).all()) for Lw in eng.layers)
        keep = ops.NORM_FUSION
        ops.NORM_FUSION = True                                   # (the default: every part)
        try:
            fused = eng.explain(ids, target=tgt)
            ops.NORM_FUSION = False
            plain = eng.explain(ids, target=tgt)
        finally:
            ops.NORM_FUSION = keep
        e_f, e_p = nmax(fused["R_tok"], ref["R_tok"]), nmax(plain["R_tok"], ref["R_tok"])
        d_fp = nmax(fused["R_tok"], plain["R_tok"])
        print(f"[K1n sparse_top={sparse_top}] vs the fp32 engine: fused {e_f:.2e}, stand-alone {e_p:.2e}; fused vs stand-alone {d_fp:.2e}")
        # one prompt set of one instance (B = 3): a few bf16 ulps of the largest token relevance; which flow is the more accurate one is a
        # statement over many prompts (tools/k1n_error_parts.py, profiles/r06_fused_flow_error_parts.txt: the fused one) and is anchored on the
        # oracle at full width in test_baseline_size_gpu.py::test_engine_bf16_full_width_batched_fused_vs_oracle
        assert torch.isfinite(fused["R_tok"]).all() and e_f < 2e-2 and e_p < 2e-2 and d_fp < 1.5e-2
        if sparse_top:
            fused = eng.explain(ids, target=tgt)                 # the default parts
            assert nmax(fused["R_tok"], ref["R_tok"]) < 2e-2
            again = eng.explain(ids, target=tgt, graph=True)
            again = eng.explain(ids, target=tgt, graph=True)
            assert torch.equal(again["R_tok"], fused["R_tok"])
            flip = eng.explain(ids.flip(0), target=tgt.flip(0), layer_relevance=True)       # prompts are independent rows of the GEMMs (same row
            assert torch.equal(flip["R_tok"].flip(0), fused["R_tok"])                        # count M: the same kernels), whatever the call's options
            assert nmax(flip["R_tok"].sum(1), flip["layer_R"][0]) < 2e-2
        del eng
    # explicit placement at the SAME row count (M = 6144): the K1n forward runs here too (round 6: lrp_gemm_res_ssq keeps each Linear's own output
    # for the stabilisers; the backward keeps its stand-alone add2 / stabiliser kernels), on the folded weights
    ref_x = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode="explicit", max_seq=S).explain(ids, target=tgt)
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.bfloat16, mode="explicit", max_seq=S)
    assert eng._norm_fused(B * S, fwd_only=True) and not eng._norm_fused(B * S)
    out_x = eng.explain(ids, target=tgt)
    one_x = eng.explain(ids[:1], target=tgt[:1])                     # M = 2048: the stand-alone forward kernels
    unf = eng_mod.LlamaLRP(cfg, W, dtype=torch.bfloat16, mode="explicit", max_seq=S, fold_norm=False).explain(ids, target=tgt)
    e_x = [nmax(out_x["R_tok"][b], ref_x["R_tok"][b]) for b in range(B)]
    e_u = [nmax(unf["R_tok"][b], ref_x["R_tok"][b]) for b in range(B)]
    gm = lambda v: float(torch.tensor(v).log().mean().exp())      # noqa: E731
    print(f"[K1n forward, explicit placement on folded weights] vs the fp32 explicit engine per prompt {[f'{x:.2e}' for x in e_x]} "
          f"(unfolded bf16 engine, stand-alone kernels: {[f'{x:.2e}' for x in e_u]}); batched vs single-prompt call {nmax(out_x['R_tok'][0], one_x['R_tok'][0]):.2e}")
    assert torch.isfinite(out_x["R_tok"]).all() and gm(e_x) < max(5e-2, 3 * gm(e_u))


@pytest.mark.parametrize("mode", ["explicit", "efficient"])
def test_llama_bf16(eng_mod, mode):
    """bf16 engine against the fp64 oracle on the bf16-rounded weights.  Efficient placement: <= 5e-2.  Explicit placement in bf16 is
    rounding noise next to the stabilisers' poles (o / (o + 1e-6) on bf16 activations): the value moves with every change of a kernel's
    rounding pattern (1.5e-2 ... 7.9e-2 over rounds 2 - 4 on this instance) and the reference's OWN arithmetic run in bf16 sits at 2.2e-2 ...
    3.8e-2 depending on the host's bf16 matmul.  One instance is one draw: the evidence is the distribution over instances in
    test_llama_bf16_explicit_seed_set (engine geometric mean 0.8 x, median 1.7 x the reference arithmetic in bf16; single instances 0.06 x ...
    6.4 x); here the same per-instance condition as there: <= 10 x the instance's own yardstick + 5e-2."""
    cfg, W, ids, fx = llama_case("d128")
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.bfloat16, mode=mode, max_seq=512)
    out = eng.explain(ids[None], target=torch.tensor([int(fx["idx"])]))
    # bf16 reference: the oracle on the bf16-rounded weights, fp64 arithmetic
    Wb = ol.cast_weights(ol.cast_weights(W, torch.bfloat16), torch.float32)
    o64 = ol.explain(cfg, Wb, ids=ids, target=int(fx["idx"]), mode="efficient", dtype=torch.float64)
    e = nmax(out["R_tok"][0], o64["R_tok"])
    bar = 5e-2
    if mode == "explicit":
        ob = ol.explain(cfg, Wb, ids=ids, target=int(fx["idx"]), mode="explicit", dtype=torch.bfloat16)
        gap = nmax(ob["R_tok"].double(), o64["R_tok"])
        bar = 10 * gap + 5e-2
        print(f"[bf16/explicit] the reference arithmetic in bf16 vs fp64 on this instance: {gap:.2e}")
    print(f"[bf16/{mode}] tok vs fp64 oracle on bf16 weights {e:.2e} (bar {bar:.2e})")
    assert torch.isfinite(out["R_tok"]).all() and e < bar


def test_llama_bf16_explicit_seed_set(eng_mod):
    """explicit placement in bf16 over eight seeded instances (H 512, 4 / 1 heads of d = 128, S = 192): engine error vs the fp64 oracle next to
    the error of the reference's own arithmetic run in bf16 (oracle, CPU) on the same instance.  Distributional bar as in the fp32 full-width
    test: geometric mean and median of the engine <= 3 x those of the reference arithmetic; no instance beyond 10 x its own yardstick."""
    cfg, _, _, _ = llama_case("d128")
    rows = []
    for seed in range(8):
        W = ol.random_weights(cfg, seed=500 + seed)
        ids = torch.randint(0, cfg["vocab"], (192,), generator=torch.Generator().manual_seed(900 + seed))
        Wb = ol.cast_weights(ol.cast_weights(W, torch.bfloat16), torch.float32)
        o64 = ol.explain(cfg, Wb, ids=ids, mode="explicit", dtype=torch.float64)
        idx = int(o64["idx"])
        ob = ol.explain(cfg, Wb, ids=ids, target=idx, mode="explicit", dtype=torch.bfloat16)
        eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.bfloat16, mode="explicit", max_seq=256)
        out = eng.explain(ids[None], target=torch.tensor([idx]))
        assert torch.isfinite(out["R_tok"]).all()
        rows.append((nmax(out["R_tok"][0], o64["R_tok"]), nmax(ob["R_tok"].double(), o64["R_tok"])))
        eng.release()
    e = torch.tensor([r[0] for r in rows]).double()
    g = torch.tensor([r[1] for r in rows]).double()
    gm = lambda t: float(t.log().mean().exp())  # noqa: E731
    print("[bf16 explicit, 8 instances] engine vs fp64 | reference arithmetic in bf16 vs fp64")
    for a, b in rows:
        print(f"    {a:.2e} | {b:.2e}  ({a / b:.2f}x)")
    print(f"    geometric mean {gm(e):.2e} | {gm(g):.2e}; median {float(e.median()):.2e} | {float(g.median()):.2e}")
    assert gm(e) <= 3 * gm(g) and float(e.median()) <= 3 * float(g.median())
    assert all(a <= 10 * b + 5e-2 for a, b in rows)


def test_conservation_large(eng_mod):
    """size-independent property at a larger shape (no oracle needed): without biases the
    efficient rules conserve relevance up to what RMSNorm/softmax absorb; the sum of token
    relevance equals the latent relevance entering the embedding and stays O(logit)."""
    cfg = dict(hidden=1024, inter=2048, n_layers=2, n_heads=8, n_kv=2, head_dim=128, vocab=1024, rope_theta=5e5, rms_eps=1e-5)
    W = ol.random_weights(cfg, seed=7)
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode="efficient", max_seq=1024)
    ids = torch.randint(0, 1024, (2, 1024), generator=torch.Generator().manual_seed(3))
    out = eng.explain(ids, layer_relevance=True)
    assert torch.isfinite(out["R_tok"]).all()
    assert nmax(out["R_tok"].sum(1), out["layer_R"][0]) < 1e-4
    assert (out["R_tok"].sum(1).abs() < 10 * out["logit"].abs() + 1).all()


@pytest.mark.parametrize("S,B", [(37, 1), (100, 3), (333, 2)])
@pytest.mark.parametrize("mode", ["explicit", "efficient"])
def test_llama_ragged_lengths(eng_mod, S, B, mode):
    """sequence lengths that are not multiples of any tile (attention tails, GEMM M tails), batches > 1"""
    cfg = dict(hidden=256, inter=512, n_layers=2, n_heads=8, n_kv=2, head_dim=32, vocab=512, rope_theta=5e5, rms_eps=1e-5)
    W = ol.random_weights(cfg, seed=302)
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode=mode, max_seq=512)
    ids = torch.randint(0, 512, (B, S), generator=torch.Generator().manual_seed(S))
    out = eng.explain(ids)
    for b in range(B):
        ref = ol.explain(cfg, W, ids=ids[b], target=int(out["idx"][b]), mode=mode, dtype=torch.float64)
        err = nmax(out["R_tok"][b], ref["R_tok"])
        if mode == "efficient":                       # no stabilisers, no poles: the north star's bar outright
            assert err < 1e-4, (S, b, err)
            continue
        # explicit: 1e-4 wherever the REFERENCE's own fp32 run resolves the instance, else 3x the reference's own fp32 gap on it
        # (tests/golden/small_cases_ref.npz: lxt.explicit's Functions composed as lxt/explicit/models/llama.py, run in the build container)
        fx = ref_case(f"llama_ragged_S{S}_b{b}")
        assert int(out["idx"][b]) == fx["idx"] and nmax(ref["R_tok"], fx["R_tok"]) < 1e-9      # same instance, same exact result
        print(f"[ragged explicit S={S} prompt {b}] engine vs exact {err:.2e} | the reference's own fp32 {fx['gap']:.2e}")
        assert err < ref_bar(fx["gap"]), (S, b, err, fx["gap"])


@pytest.mark.parametrize("mode", ["explicit", "efficient"])
def test_llama_left_padded_batch(eng_mod, mode):
    """prompts of different lengths in ONE fused call (left-padded, pad keys masked by per-row key intervals): every
    prompt equals its own un-padded oracle explanation; pad positions carry exactly zero relevance"""
    cfg = dict(hidden=256, inter=512, n_layers=3, n_heads=8, n_kv=2, head_dim=32, vocab=512, rope_theta=5e5, rms_eps=1e-5)
    W = ol.random_weights(cfg, seed=303)
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode=mode, max_seq=512)
    S, lens = 150, [150, 97, 31, 1]
    ids = torch.randint(0, 512, (len(lens), S), generator=torch.Generator().manual_seed(5))
    out = eng.explain(ids, lengths=lens)
    assert torch.isfinite(out["R_tok"]).all()
    for b, n in enumerate(lens):
        sub = ids[b, S - n:]
        single = eng.explain(sub[None], target=out["idx"][b:b + 1])                 # the same prompt alone, un-padded
        ref = ol.explain(cfg, W, ids=sub, target=int(out["idx"][b]), mode=mode, dtype=torch.float64)
        e_pad, e_one = nmax(out["R_tok"][b, S - n:], ref["R_tok"]), nmax(single["R_tok"][0], ref["R_tok"])
        e_ps = nmax(out["R_tok"][b, S - n:], single["R_tok"][0])
        assert abs(float(out["logit"][b]) - float(ref["logit"])) < 1e-4 * max(1.0, abs(float(ref["logit"])))
        assert (out["R_tok"][b, : S - n] == 0).all()
        if n == S:
            assert e_ps == 0.0                                                        # no padding: bit-identical to the plain path
        # padding must not add error beyond the un-padded evaluation's own
        assert e_ps < max(1e-5, 3 * e_one), (b, n, e_ps, e_one)
        if mode == "efficient":
            assert e_pad < 1e-4, (b, n, e_pad)
        else:                                         # explicit: against the reference's own fp32 run of the same (un-padded) prompt
            fx = ref_case(f"llama_leftpad_b{b}")
            assert int(out["idx"][b]) == fx["idx"] and nmax(ref["R_tok"], fx["R_tok"]) < 1e-9
            print(f"[left-padded explicit prompt {b}, length {n}] padded {e_pad:.2e} alone {e_one:.2e} | the reference's own fp32 {fx['gap']:.2e}")
            assert max(e_pad, e_one) < ref_bar(fx["gap"]), (b, n, e_pad, e_one, fx["gap"])


@pytest.mark.parametrize("mode", ["explicit", "efficient"])
def test_llama_dense_seed_contrastive(eng_mod, mode):
    """`logits[0,-1].backward(mask)` with a dense mask (contrastive explanation, ref docs/source/quickstart.rst:267-270):
    +1 on the arg-max logit, -1/V elsewhere; explicit mode seeds that pattern as relevance (mask * logits)"""
    cfg = dict(hidden=256, inter=512, n_layers=3, n_heads=8, n_kv=2, head_dim=32, vocab=512, rope_theta=5e5, rms_eps=1e-5)
    W = ol.random_weights(cfg, seed=311)
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode=mode, max_seq=256)
    ids = torch.randint(0, 512, (2, 96), generator=torch.Generator().manual_seed(9))
    base = eng.explain(ids)
    V = cfg["vocab"]
    mask = torch.full((2, V), -1.0 / V)
    mask[torch.arange(2), base["idx"].long().cpu()] = 1.0
    seed = mask * base["logits"].cpu() if mode == "explicit" else mask
    out = eng.explain(ids, seed=seed)
    one_hot = torch.zeros(2, V)
    one_hot[torch.arange(2), base["idx"].long().cpu()] = 1.0
    out1 = eng.explain(ids, seed=(one_hot * base["logits"].cpu() if mode == "explicit" else one_hot))
    for b in range(2):
        assert nmax(out1["R_tok"][b], base["R_tok"][b]) < 1e-5               # a one-hot seed is the target path
        ref = ol.explain(cfg, W, ids=ids[b], mode=mode, dtype=torch.float64, seed=seed[b].double())
        err = nmax(out["R_tok"][b], ref["R_tok"])
        if mode == "efficient":
            print(f"[dense seed efficient prompt {b}] engine vs oracle fp64 {err:.2e}")
            assert err < 1e-4
        else:                                         # the same contrastive seed through the reference's own Functions (fp32 vs exact)
            fx = ref_case(f"llama_dense_seed_b{b}")
            assert int(base["idx"][b]) == fx["idx"] and nmax(ref["R_tok"], fx["R_tok"]) < 1e-5    # (the seed is built from fp32 logits here)
            print(f"[dense seed explicit prompt {b}] engine vs oracle fp64 {err:.2e} | the reference's own fp32 {fx['gap']:.2e}")
            assert err < ref_bar(fx["gap"])


def test_full_width_properties_bf16(eng_mod):
    """BASELINE-size layer width (H 4096, I 14336, 32/8 heads, d 128, S 2048, bf16), 2 layers: no oracle at this
    size, so size-independent properties: finite, independence of the batch neighbours (bit for bit at equal row count), sum of token relevance ==
    latent relevance at the embedding, target override honoured."""
    cfg = dict(hidden=4096, inter=14336, n_layers=2, n_heads=32, n_kv=8, head_dim=128, vocab=4096, rope_theta=5e5, rms_eps=1e-5)
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.02).bfloat16()  # noqa: E731
    H, I, d = 4096, 14336, 128
    W = dict(embed=rn(4096, H), norm=torch.ones(H, device="cuda").bfloat16(), lm_head=rn(4096, H),
             layers=[dict(ln1=torch.ones(H, device="cuda").bfloat16(), ln2=torch.ones(H, device="cuda").bfloat16(), wq=rn(32 * d, H),
                          wk=rn(8 * d, H), wv=rn(8 * d, H), wo=rn(H, 32 * d), wg=rn(I, H), wu=rn(I, H), wd=rn(H, I)) for _ in range(2)])
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.bfloat16, mode="efficient", max_seq=2048)
    ids = torch.randint(0, 4096, (2, 2048), generator=torch.Generator().manual_seed(1))
    both = eng.explain(ids, layer_relevance=True)
    assert torch.isfinite(both["R_tok"]).all()
    # a prompt's explanation does not depend on its neighbours in the batch: bit for bit wherever the same kernels serve the call (the
    # same row count M = B S) ...
    swapped = eng.explain(ids.flip(0))
    for b in range(2):
        assert torch.equal(swapped["R_tok"][1 - b], both["R_tok"][b])
    # ... and within bf16 rounding across row counts: at M = 2048 the GEMMs against 4096-row weights split their K range over two
    # workgroups (ops.splitk_ok: half of the chip would idle otherwise), i.e. another summation order
    for b in range(2):
        one = eng.explain(ids[b:b + 1])
        assert int(one["idx"][0]) == int(both["idx"][b]) and nmax(one["R_tok"][0], both["R_tok"][b]) < 2e-2
    assert nmax(both["R_tok"].sum(1), both["layer_R"][0]) < 2e-2        # bf16 G read-out vs fp32 row sums
    forced = eng.explain(ids[:1], target=torch.tensor([7]))
    assert int(forced["idx"][0]) == 7


@pytest.mark.parametrize("mode", ["explicit", "efficient"])
def test_fused_gated_epilogues_vs_unfused_at_engine_level(eng_mod, mode):
    """full layer width, bf16, both rule placements: the explanation with the gated-MLP rules inside the GEMM epilogues (round 6: the forward
    stashes the backward's coefficients -- lrp_gemm_gated_fwd_coef / _bwd_coef, what the product runs) against the one with GEMM + element-wise
    rule kernels on the stored gate/up output.  The two differ by bf16 rounding only (the fused form rounds the coefficient once where the pair
    rounds g, u and act(g)); both are held against the fp32 engine on the same weights, and the fused form may not be the worse one."""
    import lxt_amd.ops as O
    cfg = dict(hidden=4096, inter=14336, n_layers=2, n_heads=32, n_kv=8, head_dim=128, vocab=2048, rope_theta=5e5, rms_eps=1e-5)
    g = torch.Generator(device="cuda").manual_seed(3)
    rn = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 0.02).bfloat16()  # noqa: E731
    H, I, d = 4096, 14336, 128
    W = dict(embed=rn(2048, H), norm=1 + rn(H), lm_head=rn(2048, H),
             layers=[dict(ln1=1 + rn(H), ln2=1 + rn(H), wq=rn(32 * d, H), wk=rn(8 * d, H), wv=rn(8 * d, H), wo=rn(H, 32 * d), wg=rn(I, H),
                          wu=rn(I, H), wd=rn(H, I)) for _ in range(2)])
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.bfloat16, mode=mode, max_seq=1024, sparse_top=False)
    ids = torch.randint(0, 2048, (4, 1024), generator=torch.Generator().manual_seed(2))       # M = 4096 rows: the 256 x 256 ping-pong kernel
    assert O.GATED_FUSION and eng._gated_coef(4096)
    try:
        fused = eng.explain(ids, layer_relevance=True)
        O.GATED_FUSION = False
        assert not eng._gated_coef(4096)
        plain = eng.explain(ids, layer_relevance=True)
        plain = {k: v.clone() for k, v in plain.items()}
    finally:
        O.GATED_FUSION = True
    eng.release()
    ref = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode=mode, max_seq=1024, sparse_top=False).explain(ids, target=fused["idx"], layer_relevance=True)
    assert torch.isfinite(fused["R_tok"]).all() and float(fused["R_tok"].abs().max()) > 0 and torch.equal(fused["idx"], plain["idx"])
    e_f, e_p, e_fp = nmax(fused["R_tok"], ref["R_tok"]), nmax(plain["R_tok"], ref["R_tok"]), nmax(fused["R_tok"], plain["R_tok"])
    print(f"[gated rules fused (coefficient stash) vs pair, {mode}] vs fp32 engine: fused {e_f:.2e} pair {e_p:.2e}; fused vs pair {e_fp:.2e}")
    assert nmax(fused["logits"], plain["logits"]) < 2e-2
    if mode == "efficient":
        assert e_f < max(1.25 * e_p, 5e-3) and e_fp < 1e-2 and nmax(fused["layer_R"], plain["layer_R"]) < 1e-2
    else:          # explicit placement in bf16 is pole noise (test_llama_bf16_explicit_seed_set): same order, no more
        assert e_f < max(3 * e_p, 5e-2)


@pytest.mark.parametrize("mode", ["explicit", "efficient"])
def test_top_layer_sparsity_equals_dense(eng_mod, mode):
    """evaluating the last layer's o-proj / MLP / attention rows only for the last token of each prompt
    (M = B) must give the same relevance as the dense evaluation"""
    cfg, W, ids, fx = llama_case("mid")
    ids2 = torch.stack([ids, ids.flip(0)])
    dense = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode=mode, max_seq=512, sparse_top=False).explain(ids2, layer_relevance=True)
    sparse = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode=mode, max_seq=512, sparse_top=True).explain(ids2, layer_relevance=True)
    assert torch.equal(dense["idx"], sparse["idx"])
    assert nmax(sparse["R_tok"], dense["R_tok"]) < 1e-5 and nmax(sparse["layer_R"], dense["layer_R"]) < 1e-5


def test_job_sharded_over_fake_ranks_equals_unsharded(eng_mod):
    """SURVEY 8e correctness contract on ONE device: the shards two ranks would compute (lxt_amd.dist.explain_sharded with the
    rank / world overridden, chunks of 2 prompts) concatenate to the un-sharded job bit for bit -- same kernels, same order"""
    import lxt_amd.dist as D
    cfg, W, ids, fx = llama_case("mid")
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode="efficient", max_seq=512)
    job = torch.stack([torch.roll(ids, k) for k in range(7)])                     # 7 prompts: uneven shards (4 + 3)
    whole = torch.cat([eng.explain(job[i:i + 1])["R_tok"] for i in range(7)])
    fn = lambda x: eng.explain(x)["R_tok"]                                        # noqa: E731
    parts = [D.explain_sharded(fn, job, batch=2, rank=r, world=2, gather=False) for r in (0, 1)]
    assert [p.shape[0] for p in parts] == [4, 3]
    assert torch.equal(torch.cat(parts), whole)


def test_two_rank_job_on_two_gpus(tmp_path):
    """world-2 run of the REAL engine (gloo rendezvous, one rank per visible GPU): weight broadcast of the flat buffer, local W^T
    rebuild, sharded explanations, per-job all-gather == rank 0's own un-sharded result.  Skipped on 1-GPU boxes."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 visible GPUs")
    import os, socket, subprocess, sys, textwrap
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys, torch
        sys.path.insert(0, %r)
        import lxt_amd.dist as D, lxt_amd.engine as E
        import torch.distributed as dist
        from tests.util import llama_case
        rank, world, local = D.init()
        cfg, W, ids, fx = llama_case("mid")
        if rank != 0:
            W = {k: (torch.zeros_like(v) if torch.is_tensor(v) else [{kk: torch.zeros_like(vv) for kk, vv in L.items()} for L in v]) for k, v in W.items()}
        eng = E.LlamaLRP(cfg, W, dtype=torch.float32, mode="efficient", max_seq=512, device=f"cuda:{local}")
        D.broadcast_weights([eng.flat], src=0)
        eng.build_transposes()
        job = torch.stack([torch.roll(ids, k) for k in range(5)]).to(f"cuda:{local}")
        R = D.explain_sharded(lambda x: eng.explain(x)["R_tok"], job, batch=2)
        if rank == 0:
            whole = torch.cat([eng.explain(job[i:i + 1])["R_tok"] for i in range(5)])
            assert torch.equal(R, whole), "sharded != single"
        dist.barrier()
        open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ok%%d" %% rank), "w").write("ok")
    """) % root)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists(), r.stdout[-2000:] + r.stderr[-2000:]


def test_single_rank_job_through_rccl(tmp_path):
    """the job's collectives on RCCL with the ONE GPU a test box has: a one-rank `nccl` process group, the short cuts for world == 1 switched
    off (dist.SINGLE_RANK_COLLECTIVES): broadcast of the engine's flat weight buffer, all-gather of the replica checksums (int64 on the
    device), the rank-tagged gather-order check and the job's own all-gather of fp32 relevances -- the same calls, buffers and dtypes the
    8-rank job issues, through librccl; result equal to the un-sharded explanation bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    import os, socket, subprocess, sys, textwrap
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    script = tmp_path / "w1.py"
    script.write_text(textwrap.dedent("""
        import os, sys, torch
        sys.path.insert(0, %r)
        import lxt_amd.dist as D, lxt_amd.engine as E
        import torch.distributed as dist
        from tests.util import llama_case
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        assert dist.get_backend() == "nccl"
        D.SINGLE_RANK_COLLECTIVES = True
        cfg, W, ids, fx = llama_case("mid")
        eng = E.LlamaLRP(cfg, W, dtype=torch.bfloat16, mode="efficient", max_seq=512, device="cuda:0")
        before = D.checksum_list([eng.flat])
        D.broadcast_weights([eng.flat], src=0)
        sums = D.check_replicas([eng.flat])
        assert sums == [before], (sums, before)
        D.check_gather_order(7, ids.shape[0], torch.device("cuda:0"))
        job = torch.stack([torch.roll(ids, k) for k in range(5)]).cuda()
        R = D.explain_sharded(lambda x: eng.explain(x)["R_tok"], job, batch=2)
        whole = torch.cat([eng.explain(job[i:i + 2])["R_tok"] for i in range(0, 5, 2)])
        assert R.shape == whole.shape and torch.equal(R, whole), "gathered != local"
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        print("rccl ok", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
    """) % root)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0"))
    print(r.stdout[-300:])
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_llama_arena_reuse_and_graph_replay(eng_mod, dtype):
    """the workspace arena is reused across calls of different (B, S) without cross-talk (results equal a fresh engine's, bit for bit),
    and graph=True (the ~1500 launches of an explanation captured once as a hipGraph) reproduces the eager launches bit for bit -- also
    after the static inputs are overwritten with another batch, with a given target, and with per-layer relevance"""
    cfg, W, ids, fx = llama_case("d128" if dtype == torch.bfloat16 else "mid")
    mk = lambda: eng_mod.LlamaLRP(cfg, W, dtype=dtype, mode="explicit", max_seq=512)      # noqa: E731
    g = torch.Generator().manual_seed(9)
    S = ids.shape[0]
    a = torch.stack([ids, torch.randint(0, cfg["vocab"], (S,), generator=g)])
    b = torch.randint(0, cfg["vocab"], (1, S // 2), generator=g)
    c = torch.randint(0, cfg["vocab"], (2, S), generator=g)
    eng = mk()
    r_a1 = eng.explain(a, layer_relevance=True)
    r_b = eng.explain(b)
    r_a2 = eng.explain(a, layer_relevance=True)                 # same buffers, after a smaller problem used them
    fresh_a, fresh_b = mk().explain(a, layer_relevance=True), mk().explain(b)
    for r in (r_a1, r_a2):
        assert torch.equal(r["R_tok"], fresh_a["R_tok"]) and torch.equal(r["layer_R"], fresh_a["layer_R"]) and torch.equal(r["logits"], fresh_a["logits"])
    assert torch.equal(r_b["R_tok"], fresh_b["R_tok"])
    assert eng._arena.nbytes() > 0
    # hipGraph replay
    g1 = eng.explain(a, layer_relevance=True, graph=True)
    assert torch.equal(g1["R_tok"], fresh_a["R_tok"]) and torch.equal(g1["idx"], fresh_a["idx"]) and torch.equal(g1["layer_R"], fresh_a["layer_R"])
    eager_c = mk().explain(c, layer_relevance=True)
    g2 = eng.explain(c, layer_relevance=True, graph=True)        # replay of the captured graph on new ids
    assert torch.equal(g2["R_tok"], eager_c["R_tok"]) and torch.equal(g2["idx"], eager_c["idx"])
    tgt = (eager_c["idx"].cpu().long() + 1) % cfg["vocab"]
    g3 = eng.explain(c, target=tgt, graph=True)
    e3 = mk().explain(c, target=tgt)
    assert torch.equal(g3["R_tok"], e3["R_tok"]) and torch.equal(g3["idx"].cpu().long(), tgt)
    with pytest.raises(ValueError):
        eng.explain(c, lengths=[S, S - 3], graph=True)


def test_llama_graph_survives_arena_growth(eng_mod):
    """a graph captured at a SMALL (B, S) must not be replayed into freed memory after a later, larger call re-allocated
    the arena's buffers: graph small -> eager large (the arena grows, its old blocks return to the allocator and are overwritten by
    fresh tensors) -> replay small.  The stale graph is dropped and re-captured (arena generation in the graph record)."""
    cfg, W, ids, fx = llama_case("mid")
    g = torch.Generator().manual_seed(11)
    S = ids.shape[0]
    small = torch.randint(0, cfg["vocab"], (1, S // 2), generator=g)
    big = torch.randint(0, cfg["vocab"], (3, S), generator=g)
    eng = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode="efficient", max_seq=512)
    ref_small = eng_mod.LlamaLRP(cfg, W, dtype=torch.float32, mode="efficient", max_seq=512).explain(small)
    g1 = eng.explain(small, graph=True)
    assert torch.equal(g1["R_tok"], ref_small["R_tok"])
    gen0 = eng._arena.gen
    r_big = eng.explain(big)                                  # grows every arena buffer
    assert eng._arena.gen > gen0
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(16)]      # land on the freed blocks
    g2 = eng.explain(small, graph=True)
    assert torch.equal(g2["R_tok"], ref_small["R_tok"]) and torch.isfinite(g2["R_tok"]).all()
    assert torch.isfinite(r_big["R_tok"]).all()
    del junk
