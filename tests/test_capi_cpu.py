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
    assert L.lib.lrp_version() == 1 and L.lib.lrp_build_arch() == b"gfx950"


def test_argument_validation_without_gpu():
    import lxt_amd._lib as L
    # rejected before any launch: null pointers / bad dtype / misaligned K
    assert L.lib.lrp_gemm_nt(None, None, None, None, 4, 4, 8, 8, 8, 4, 1, 0, 0, 0, 0, 0, None) == -1
    assert L.lib.lrp_gemm_nt(16, 16, 16, None, 4, 4, 6, 8, 8, 4, 1, 0, 0, 0, 0, 0, None) == -2
    assert L.lib.lrp_eps_scale(None, None, None, 10, 1.0, 1e-6, 0, 0, None) == -1
    assert L.lib.lrp_attn_fwd(16, 16, 16, 16, 16, 1, 8, 3, 2, 64, 64, 64, 8, 64, 1.0, 1, 0, 0, None, None, 1, None) == -1  # Hq % Hkv


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
