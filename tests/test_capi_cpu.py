"""CPU: the C-ABI library loads and exports every symbol include/lrp_hip.h declares; the
binding fails loudly (no fallback) when handed CPU tensors."""
import ctypes
import os

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_header_symbols_exported():
    import lxt_amd._lib as L
    decls = L.parse_header()
    assert len(decls) >= 30
    raw = ctypes.CDLL(L.LIB_PATH)
    for name in decls:
        assert hasattr(raw, name), f"{name} declared in lrp_hip.h but not exported"
    assert L.lib.lrp_version() == 8 and L.lib.lrp_build_arch() == b"gfx950"


def test_argument_validation_without_gpu():
    import lxt_amd._lib as L
    # rejected before any launch: null pointers / bad dtype / misaligned K
    assert L.lib.lrp_gemm_nt(None, None, None, None, 4, 4, 8, 8, 8, 4, 1, 0, 0, 0, 0, 0, None) == -1
    assert L.lib.lrp_gemm_nt(16, 16, 16, None, 4, 4, 6, 8, 8, 4, 1, 0, 0, 0, 0, 0, None) == -2
    assert L.lib.lrp_eps_scale(None, None, None, 10, 1.0, 1e-6, 0, 0, None) == -1
    assert L.lib.lrp_attn_fwd(16, 16, 16, 16, 16, 16, 1, 8, 3, 2, 64, 64, 64, 64, 8, 64, 1.0, 1, 0, 0, None, None, 1, None) == -1  # Hq % Hkv


def test_round5_host_side_queries_and_layout_rules():
    """the dispatch questions of the round-5 entry points are host code (no launch): K1n applicability, the stream forward's K splits / workspace /
    ticket words, the dQ-with-D kernel's coverage, argument checks of the new calls -- and the engine's row-pitch rules"""
    import lxt_amd._lib as L
    lib, BF16, F32 = L.lib, L.BF16, L.F32
    # K1n: bf16 only, N % 256 == 0, >= 190 tiles of 256 x 256
    assert lib.lrp_gemm_norm_fused_ok(8192, 4096, 4096, 4096, 4096, 0, BF16) == 1
    assert lib.lrp_gemm_norm_fused_ok(8192, 4096, 28672, 28736, 4224, 1, BF16) == 1
    assert lib.lrp_gemm_norm_fused_ok(8192, 4096 + 64, 4096, 4096, 4096, 0, BF16) == 0          # N % 256
    assert lib.lrp_gemm_norm_fused_ok(2048, 4096, 4096, 4096, 4096, 0, BF16) == 0               # 128 tiles
    assert lib.lrp_gemm_norm_fused_ok(8192, 4096, 4096, 4096, 4096, 0, F32) == 0
    assert lib.lrp_gemm_res_ssq(None, None, None, None, None, 8, 256, 128, 128, 128, 256, 256, 8, None, 0, BF16, None) == -1
    assert lib.lrp_rms_rstd(None, 4, 8, 8, 256, 1e-5, None, None) == -1
    # stream forward: wide weights run full K (1 split), narrow ones 2 ... 8 splits with whole 8-tile rings; ws = splits * M * N * 4 bytes
    assert lib.lrp_linear_stream_fwd_splits(4, 14336, 4096) == 1 and lib.lrp_linear_stream_fwd_ws(4, 14336, 4096) == 0
    assert lib.lrp_linear_stream_fwd_splits(4, 4096, 14336) == 4 and lib.lrp_linear_stream_fwd_ws(4, 4096, 14336) == 4 * 4 * 4096 * 4
    assert lib.lrp_linear_stream_fwd_tickets(4, 4096, 14336) == 64 and lib.lrp_linear_stream_fwd_tickets(4, 14336, 4096) == 0
    assert lib.lrp_linear_stream_fwd_splits(4, 8192, 2048) == 2 and lib.lrp_linear_stream_fwd_splits(4, 1024, 4096) == 0
    assert lib.lrp_linear_stream_fwd_splits(200, 4096, 14336) == 0                                # M > 128
    assert lib.lrp_linear_stream_ok(4, 4096, 14336, 14336, 14400) == 1 and lib.lrp_linear_stream_ok(4, 1024, 4096, 4096, 4096) == 0
    # dQ with D (and RoPE's backward): the bf16 32x32 kernels of head dims 64 / 96 / 128
    assert [lib.lrp_attn_bwd_dq_d_ok(BF16, d) for d in (64, 96, 128, 256, 32)] == [1, 1, 1, 0, 0] and lib.lrp_attn_bwd_dq_d_ok(F32, 128) == 0
    assert lib.lrp_gqa_reduce_rope(None, None, 8, 8, 2, 2, 64, 256, 128, None, None, BF16, None) == -1
    # RoPE in the QKV forward's epilogue: heads of 128, M and N multiples of 256, seq a multiple of 16
    assert lib.lrp_gemm_nt_rs_rope_ok(8192, 6144, 4096, 4096, 4224, 6144, 2048, 5120, 128, BF16) == 1
    assert lib.lrp_gemm_nt_rs_rope_ok(8192, 6144, 4096, 4096, 4224, 6144, 2048, 5120, 64, BF16) == 0
    assert lib.lrp_gemm_nt_rs_rope_ok(8200, 6144, 4096, 4096, 4224, 6144, 2050, 5120, 128, BF16) == 0
    assert lib.lrp_gemm_nt_rs_rope_ok(8192, 6144, 4096, 4096, 4224, 6144, 2048, 5000, 128, BF16) == 0
    assert lib.lrp_gemm_nt_rs_rope(None, None, None, None, None, None, 256, 256, 128, 128, 128, 256, 256, 256, 128, BF16, None) == -1
    # the fused gated-MLP pair (coefficient stash): both launches must be >= 190-tile bf16 problems
    assert lib.lrp_gemm_gated_coef_ok(8192, 14336, 4096, 4096, 4224, 4096, 14400, 0, BF16) == 1
    assert lib.lrp_gemm_gated_coef_ok(2048, 14336, 4096, 4096, 4224, 4096, 14400, 0, BF16) == 1        # 8 x 56 = 448 tiles in the backward
    assert lib.lrp_gemm_gated_coef_ok(256, 14336, 4096, 4096, 4224, 4096, 14400, 0, BF16) == 0         # 56 tiles
    assert lib.lrp_gemm_gated_coef_ok(8192, 14336, 4096, 4096, 4224, 4096, 14400, 2, BF16) == 0        # erf-GELU: no fused form
    assert lib.lrp_gemm_gated_coef_ok(8192, 14336, 4096, 4096, 4224, 4096, 14400, 0, F32) == 0
    assert lib.lrp_gemm_gated_fwd_coef(None, None, None, None, None, 8, 32, 128, 128, 128, 64, 32, 1e-10, 0.0, 0, BF16, None) == -1
    assert lib.lrp_gemm_gated_bwd_coef(None, None, None, None, 8, 32, 128, 128, 32, 64, 64, BF16, None) == -1
    # row-pitch rules (pure functions)
    import lxt_amd.engine as E
    assert E.weight_pitch_pad(4096, 2, 28672) == 128 and E.weight_pitch_pad(4096, 2, 2048) == 0          # 8-KiB pitch: padded when the weight exceeds the caches
    assert E.weight_pitch_pad(2560, 2, 20480) == 128 and E.weight_pitch_pad(1152, 2, 4352) == 0          # 5 KiB (Gemma-3): padded; SigLIP: small, odd pitch
    assert E.pitch_pad(14336, 2) == 64 and E.pitch_pad(28672, 2) == 64 and E.pitch_pad(4096, 2) == 0
    import lxt_amd.ops as ops
    keep = ops.NORM_FUSION
    try:
        ops.NORM_FUSION = frozenset({"fwd"})
        assert ops.norm_fusion_part("fwd") and not ops.norm_fusion_part("bwd_gu")
        ops.NORM_FUSION = True
        assert ops.norm_fusion_part("bwd_gu")
        ops.NORM_FUSION = False
        assert not ops.norm_fusion_part("fwd")
    finally:
        ops.NORM_FUSION = keep


def test_no_cpu_fallback():
    import lxt_amd.ops as ops
    a = torch.randn(4, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm_nt(a, a)
    if not torch.cuda.is_available():
        import lxt_amd.engine as E
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            E.LlamaLRP(dict(hidden=8, inter=8, n_layers=0, n_heads=1, n_kv=1, head_dim=8, vocab=8, rope_theta=1e4, rms_eps=1e-5),
                       dict(embed=a, norm=a[0], lm_head=a, layers=[]))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "lrp-explains-transformers_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_mask_plan_reduces_hf_masks_to_row_intervals():
    """host logic of the drop-in attention (no kernel call): HF's 4-D masks -> (causal, window, per-row [lo, hi))"""
    import types
    import torch
    from lxt_amd.efficient.patches import _mask_plan
    S, mod = 12, types.SimpleNamespace(is_causal=True)
    i = torch.arange(S)
    causal = i[None, :] <= i[:, None]
    neg = torch.finfo(torch.float32).min
    assert _mask_plan(None, S, mod) == (True, 0, None)
    assert _mask_plan(None, S, types.SimpleNamespace(is_causal=False)) == (False, 0, None)
    add = torch.where(causal, 0.0, neg)[None, None]
    assert _mask_plan(add, S, mod) == (True, 0, None)                                  # additive causal -> structural fast path
    assert _mask_plan(torch.ones(1, 1, S, S, dtype=torch.bool), S, mod) == (False, 0, None)
    slide = causal & (i[None, :] > i[:, None] - 4)
    assert _mask_plan(slide[None, None], S, mod, window=4) == (True, 4, None)
    # left padding (3 pads) on top of causal: rows 0..2 empty, the others [3, i+1)
    pad = causal.clone()
    pad[:, :3] = False
    c, w, iv = _mask_plan(pad[None, None].clone(), S, mod)
    assert c is True and w == 0 and iv[0].dtype == torch.int32
    assert iv[0][0].tolist() == [0, 0, 0] + [3] * (S - 3) and iv[1][0].tolist() == [0, 0, 0] + list(range(4, S + 1))
    # bidirectional block [4, 8) inside a causal prompt: not causal-bounded
    blk = causal.clone()
    blk[4:8, 4:8] = True
    c, w, iv = _mask_plan(blk[None, None].clone(), S, mod)
    assert c is False and iv[1][0, 4:8].tolist() == [8] * 4
    # a row with a hole is refused loudly
    hole = causal.clone()
    hole[6, 2] = False
    with pytest.raises(NotImplementedError, match="non-contiguous"):
        _mask_plan(hole[None, None].clone(), S, mod)


def test_patched_torch_nn_classes_leave_foreign_modules_alone():
    """after monkey_patch(modeling_llama) / (modeling_bert) the class-level nn.Linear / nn.LayerNorm patches must only act on
    instances of an explained model: a CPU nn.Linear or nn.LayerNorm elsewhere in the process still runs torch's own forward
    (bit-identical), keeps its parameter gradients, and a model class of the patched module adopts its instances on the
    first call.  The reference never patches nn.Linear at all (lxt/efficient/models/llama.py:9-14).  One fresh process."""
    import subprocess
    import sys
    code = r'''
import sys, warnings, torch
sys.path.insert(0, %r)
warnings.simplefilter("ignore")
from torch import nn
import torch.nn.functional as F
from transformers.models.llama import modeling_llama
from transformers.models.bert import modeling_bert
from lxt_amd.efficient import monkey_patch, adopt
lin, ln = nn.Linear(8, 6), nn.LayerNorm(8)
x = torch.randn(3, 8)
y0, n0 = lin(x).detach().clone(), ln(x).detach().clone()
monkey_patch(modeling_llama)
monkey_patch(modeling_bert)
assert nn.Linear.forward.__module__ == "lxt_amd.efficient.patches"
y1 = lin(x)
assert torch.equal(y1, y0) and torch.equal(ln(x), n0)
y1.sum().backward()
assert lin.weight.grad is not None and torch.equal(lin.weight.grad, x.sum(0)[None].expand(6, 8))
# a model class defined in the patched module adopts its nn.Linear instances on the first call
cfg = modeling_llama.LlamaConfig(hidden_size=16, intermediate_size=32, num_hidden_layers=1, num_attention_heads=2,
                                 num_key_value_heads=1, vocab_size=32)
m = modeling_llama.LlamaForCausalLM(cfg)
for p_ in m.parameters():
    p_.requires_grad_(False)
assert not m.lm_head.__dict__.get("_lrp_owned", False)
try:
    m(input_ids=torch.zeros(1, 4, dtype=torch.long))
    raise SystemExit("an adopted model on the CPU must fail loudly (no CPU fallback)")
except RuntimeError as e:
    assert "no CPU fallback" in str(e), e
assert m.lm_head.__dict__["_lrp_owned"] and m.model.layers[0].mlp.down_proj.__dict__["_lrp_owned"]
assert not lin.__dict__.get("_lrp_owned", False)
print("SCOPED-OK")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SCOPED-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_rope_scaling_tables_match_hf():
    """oracle.llama.rope_inv_freq restates HF's static rope initialisers (Llama-3.1/3.2 use rope_type='llama3'); the engine
    takes HF's own frequencies through config_from_hf.  Both against transformers' ROPE_INIT_FUNCTIONS on the CPU."""
    from transformers import LlamaConfig
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    from oracle import llama as ol
    import lxt_amd.engine as E
    rp = dict(rope_type="llama3", rope_theta=500000.0, factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
              original_max_position_embeddings=8192)
    hc = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                     head_dim=128, vocab_size=64, rope_parameters=rp)
    ref, att = ROPE_INIT_FUNCTIONS["llama3"](hc, "cpu")
    ocfg = ol.config_from_hf(hc)
    inv, oatt = ol.rope_inv_freq(ocfg)
    assert oatt == att == 1.0 and torch.equal(inv, ref)
    assert not torch.equal(inv, ol.rope_inv_freq(dict(ocfg, rope_scaling=None))[0])          # the scaling really changes the table
    ecfg = E.config_from_hf(hc)
    assert torch.equal(ecfg["inv_freq"], ref) and ecfg["attention_scaling"] == 1.0
    lin = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                      head_dim=128, vocab_size=64, rope_parameters=dict(rope_type="linear", rope_theta=10000.0, factor=4.0))
    assert torch.equal(ol.rope_inv_freq(ol.config_from_hf(lin))[0], ROPE_INIT_FUNCTIONS["linear"](lin, "cpu")[0])
    # unsupported configurations are refused loudly instead of being silently ignored
    for bad in (dict(attention_bias=True), dict(mlp_bias=True),
                dict(rope_parameters=dict(rope_type="dynamic", rope_theta=10000.0, factor=2.0))):
        kw = dict(hidden_size=64, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, vocab_size=8)
        kw.update(bad)
        with pytest.raises(NotImplementedError):
            E.config_from_hf(LlamaConfig(**kw))
    from transformers import Qwen2Config
    with pytest.raises(NotImplementedError, match="model_type"):
        E.config_from_hf(Qwen2Config(hidden_size=64, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, vocab_size=8))


def test_rope_kind_reads_legacy_rope_scaling():
    """on transformers 4.x a Llama-3.1 config has no `rope_parameters`, only rope_theta + rope_scaling = {'rope_type' (or the
    legacy 'type'): 'llama3', ...}; the scaling must not be dropped silently, and an uninterpretable dict must be refused"""
    import types
    import lxt_amd.engine as E
    ns = types.SimpleNamespace
    assert E.rope_kind(ns(rope_parameters=dict(rope_type="llama3", rope_theta=5e5))) == "llama3"
    assert E.rope_kind(ns(rope_theta=5e5, rope_scaling=dict(rope_type="llama3", factor=8.0))) == "llama3"
    assert E.rope_kind(ns(rope_theta=5e5, rope_scaling=dict(type="linear", factor=2.0))) == "linear"
    assert E.rope_kind(ns(rope_theta=1e4, rope_scaling=None)) == "default" and E.rope_kind(ns(rope_theta=1e4)) == "default"
    with pytest.raises(NotImplementedError):
        E.rope_kind(ns(rope_theta=5e5, rope_scaling={"factor": 2.0}))
    # a sequence-length dependent type arriving through the legacy key is refused by config_from_hf like any other
    with pytest.raises(NotImplementedError):
        E.config_from_hf(ns(model_type="llama", hidden_size=64, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                            num_key_value_heads=2, vocab_size=8, rms_norm_eps=1e-5, rope_theta=1e4, rope_scaling=dict(type="dynamic", factor=2.0)))


def test_bert_engine_refuses_to_run_without_a_device():
    """BertLRP (like LlamaLRP) has no CPU path: constructing it without a HIP device raises"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    from lxt_amd.engine_bert import BertLRP
    with pytest.raises(RuntimeError):
        BertLRP(dict(hidden=8, inter=16, n_layers=0, n_heads=2, ln_eps=1e-12, act="gelu", labels=2), {})


def test_explicit_bert_registers_a_mask_function():
    """transformers builds the attention mask per `_attn_implementation`; the explicit BERT wiring registers its own
    attention function, so it must register a mask builder under the same name -- otherwise padded batches get attention_mask=None.
    CPU plumbing check with a probe attention function (the real one needs the HIP library)."""
    import lxt_amd.explicit.models.bert as xb
    from transformers.models.bert.modeling_bert import eager_attention_forward
    from tests.golden.hf_models import build_bert
    model = build_bert(seed=0, attn="eager").eval()
    seen = {}

    def probe(module, query, key, value, attention_mask=None, **kw):
        seen["mask"] = attention_mask
        return eager_attention_forward(module, query, key, value, attention_mask, **kw)

    xb.register_interfaces(probe)
    try:
        ids = torch.randint(0, 1000, (2, 16), generator=torch.Generator().manual_seed(0))
        am = torch.ones(2, 16, dtype=torch.long)
        am[1, 10:] = 0
        with torch.no_grad():
            ref = model(input_ids=ids, attention_mask=am).logits
            model.config._attn_implementation = xb.ATTN_NAME
            got = model(input_ids=ids, attention_mask=am).logits
        m = seen["mask"]
        assert m is not None and tuple(m.shape) == (2, 1, 16, 16)
        assert float(m[1, 0, 0, 10:].max()) < -1e30 and float(m[0].abs().max()) == 0.0 and float(m[1, 0, :, :10].abs().max()) == 0.0
        assert torch.allclose(ref, got, atol=1e-6)
    finally:
        xb.register_interfaces()           # restore the real attention function under the name


def test_host_dispatch_rules_round3():
    """host-side logic added in round 3, none of which launches a kernel: the row-pitch padding rule, the split-K dispatch predicate,
    padded arena buffers, the gate/up interleave map, and the Gemma-3 config reader's refusals"""
    import lxt_amd.engine as E
    import lxt_amd.ops as O
    # pitch: 128 bytes of padding exactly for the long-K operands whose pitch is a multiple of 4 KiB
    assert E.pitch_pad(14336, 2) == 64 and E.pitch_pad(28672, 2) == 64 and E.pitch_pad(14336, 4) == 32
    assert E.pitch_pad(4096, 2) == 0 and E.pitch_pad(6144, 2) == 0 and E.pitch_pad(10240 + 64, 2) == 0 and E.pitch_pad(10240, 2) == 64
    # split-K: always for M <= 256 rows; for more rows only when <= 128 tiles of 256 x 256 and a K loop of >= 8 tiles per split remains
    assert O.splitk_ok(1, 128256, 4096) and O.splitk_ok(256, 4096, 128)
    assert O.splitk_ok(2048, 4096, 4096) and O.splitk_ok(2048, 4096, 28672) and O.splitk_ok(512, 4096, 4096)
    assert not O.splitk_ok(2048, 6144, 4096) and not O.splitk_ok(8192, 4096, 4096) and not O.splitk_ok(2048, 4096, 512)
    # 1 ... 1.5 rounds of the chip with a long K loop (Gemma-3-4B down projection): two K splits; not with a short one, not at 2 rounds
    assert O.splitk_ok(8192, 2560, 10240) and not O.splitk_ok(8192, 2560, 4096) and not O.splitk_ok(8192, 4096, 14336)
    # arena: a padded 2-D buffer is a [rows, cols] view with the padded pitch; the same tag is reused, a larger request grows it
    ar = E.LlamaLRP._Arena(torch.device("cpu"))
    a = ar.get("m", (8, 14336), torch.bfloat16, pad=64)
    assert a.shape == (8, 14336) and a.stride(0) == 14400 and ar.get("m", (8, 14336), torch.bfloat16, pad=64).data_ptr() == a.data_ptr()
    z = ar.get("m", (4, 14336), torch.bfloat16, zero=True, pad=64)
    assert z.data_ptr() == a.data_ptr() and float(z.abs().sum()) == 0.0
    assert ar.get("m", (16, 14336), torch.bfloat16, pad=64).data_ptr() != a.data_ptr()
    assert ar.get("x", (3, 5), torch.float32).is_contiguous()
    # gate/up interleave: blocks of 64 rows = [32 gate | 32 up]
    wg, wu = torch.arange(64.).view(64, 1).repeat(1, 2), -torch.arange(64.).view(64, 1).repeat(1, 2)
    il = O.interleave_gate_up(wg, wu)
    assert torch.equal(il[:32], wg[:32]) and torch.equal(il[32:64], wu[:32]) and torch.equal(il[64:96], wg[32:]) and torch.equal(il[96:], wu[32:])
    with pytest.raises(ValueError):
        O.interleave_gate_up(torch.zeros(48, 2), torch.zeros(48, 2))
    # Gemma-3 config reader: the supported text config is read, unsupported features are refused loudly
    import lxt_amd.engine_gemma3 as G
    from transformers import Gemma3TextConfig, LlamaConfig
    ok = Gemma3TextConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                          head_dim=16, sliding_window=8, layer_types=["sliding_attention", "full_attention"], query_pre_attn_scalar=16)
    c = G.config_from_hf(ok)
    assert c["act"] == "gelu_tanh" and c["window"] == 8 and abs(c["scale"] - 0.25) < 1e-12 and set(c["rope"]) == {"sliding_attention", "full_attention"}
    assert c["rope"]["full_attention"][0].shape == (8,) and abs(c["embed_scale"] - 32 ** 0.5) < 1e-12
    bad = Gemma3TextConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                           head_dim=16, layer_types=["full_attention"], final_logit_softcapping=30.0)
    with pytest.raises(NotImplementedError):
        G.config_from_hf(bad)
    with pytest.raises(NotImplementedError):
        G.config_from_hf(LlamaConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2))
    with pytest.raises(RuntimeError):
        G.Gemma3LRP(c, {})


def test_header_is_plain_c_and_gemma3_weight_views():
    """(1) include/lrp_hip.h is what a C caller binds: it compiles as C99 and as C++ on its own (no torch / HIP types in the signatures);
    (2) Gemma3LRP's weight reader finds the text tower in both HF model classes (no copies, tied head detected)"""
    import shutil
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    hdr = os.path.join(root, "include", "lrp_hip.h")
    if shutil.which("gcc"):
        for lang, std in (("c", "-std=c99"), ("c++", "-std=c++11")):
            r = subprocess.run(["gcc", "-fsyntax-only", "-x", lang, std, "-Wall", "-Werror", hdr], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
    import re
    code = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)     # declarations only (the comments cite torch / HIP names)
    assert "torch" not in code and "hipStream_t" not in code and "#include <hip" not in code     # streams cross the boundary as void*
    import lxt_amd.engine_gemma3 as G
    from tests.golden.hf_models import build_gemma3, build_gemma3_mm
    m1 = build_gemma3(seed=3)
    cfg, W = G.weights_from_hf(m1)
    assert len(W["layers"]) == 4 and W["layers"][0]["wq"].data_ptr() == m1.model.layers[0].self_attn.q_proj.weight.data_ptr()
    assert cfg["layer_types"][-1] == "full_attention" and W["lm_head"].data_ptr() != W["embed"].data_ptr()      # untied in this fixture
    m2 = build_gemma3_mm(seed=11)
    cfg2, W2 = G.weights_from_hf(m2)
    assert len(W2["layers"]) == 3 and cfg2["hidden"] == 64 and W2["lm_head"].data_ptr() == W2["embed"].data_ptr()   # tied (HF default)
    assert W2["layers"][0]["qn"].shape == (32,) and W2["layers"][0]["ln_pff"].shape == (64,)
