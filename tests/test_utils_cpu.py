"""lxt_amd.utils (presentation helpers, ref lxt/utils.py): host logic only"""
import pytest


def test_clean_tokens_schemes():
    from lxt_amd.utils import clean_tokens
    assert clean_tokens(["\u2581The", "\u2581cat", "s", "\u2581cost", "\u2581$5"]) == [" The", " cat", "s", " cost", " \\$5"]
    assert clean_tokens(["The", "\u0120quick", "\u0120fox_1"]) == ["The", " quick", " fox\\_1"]
    assert clean_tokens(["un", "##believ", "##able", "story"]) == ["un", "believ", "able", " story"]
    with pytest.raises(ValueError, match="not recognized"):
        clean_tokens(["plain", "words"])


def test_heatmaps(tmp_path):
    from lxt_amd import utils
    words, rel = [" a", " b", " c&d"], [-1.0, 0.0, 0.5]
    assert utils._colour(0.0) == (255, 255, 255) and utils._colour(1.0) == (255, 0, 0) and utils._colour(-1.0) == (0, 0, 255)
    tex = utils._generate_latex(words, rel)
    assert tex.count("\\colorbox[RGB]") == 3 and "{0,0,255}" in tex and "{255,128,128}" in tex
    out = utils.pdf_heatmap(words, rel, path=str(tmp_path / "h.pdf"), backend="xelatex")
    assert out.endswith((".pdf", ".tex"))                       # .tex is kept when no LaTeX is installed
    page = utils.html_heatmap(["x\\_y", " <b>"], [0.25, -0.25], path=str(tmp_path / "h.html"))
    assert "x_y" in page and "&lt;b&gt;" in page and (tmp_path / "h.html").exists()
    with pytest.raises(AssertionError, match="normalized"):
        utils.pdf_heatmap(words, [2.0, 0.0, 0.0], path=str(tmp_path / "bad.pdf"))
    with pytest.raises(AssertionError, match="same"):
        utils.html_heatmap(words, [0.0])


def test_conservation_check_mode():
    """lxt.explicit.check.conservation_check: the decorator every rule backward carries (host logic; a toy Function stands in
    for the HIP-backed rules, which need a device)"""
    import torch
    from torch.autograd import Function
    from lxt_amd.explicit.functional import conservation_check_wrap, CONSERVATION_CHECK_FLAG
    from lxt_amd.explicit.check import conservation_check
    import lxt_amd.explicit.functional as lf

    class toy(Function):
        @staticmethod
        def forward(ctx, a, b):
            return a + b

        @staticmethod
        @conservation_check_wrap
        def backward(ctx, R):
            return 0.25 * R, 0.75 * R

    a, b = torch.ones(3, requires_grad=True), torch.ones(3, requires_grad=True)
    toy.apply(a, b).backward(torch.tensor([1.0, 2.0, 3.0]))
    assert torch.allclose(a.grad, torch.tensor([0.25, 0.5, 0.75])) and torch.allclose(b.grad, torch.tensor([0.75, 1.5, 2.25]))
    a.grad = b.grad = None
    with conservation_check():
        assert CONSERVATION_CHECK_FLAG[0]
        toy.apply(a, b).backward(torch.tensor([1.0, 2.0, 3.0]))
    assert not CONSERVATION_CHECK_FLAG[0]
    assert torch.allclose(a.grad, torch.full((3,), 1.0)) and torch.allclose(b.grad, torch.full((3,), 1.0))     # 6 / 6 elements
    # every rule Function of the package carries the decorator
    for name in ("linear_epsilon_fn", "matmul_fn", "softmax_fn", "add2_tensors_fn", "mul2_fn", "rms_norm_identity_fn",
                 "layer_norm_grad_fn", "mean_fn", "normalize_identity_fn"):
        assert getattr(lf, name).backward.__name__ == "backward"


def test_monkey_patch_zennit_fails_loudly():
    from lxt_amd.efficient import monkey_patch_zennit
    with pytest.raises(NotImplementedError, match="GammaComposite"):
        monkey_patch_zennit()


def test_gemma3_config_rope_under_both_transformers_schemas():
    """engine_gemma3.config_from_hf: per-layer-type rotary frequencies from transformers-5 `rope_parameters` AND from the 4.x schema
    (rope_theta / rope_local_base_freq / rope_scaling) -- equal tables, linear scaling = HF's own initialiser, dynamic types refused loudly"""
    import pytest
    import torch
    from transformers import Gemma3TextConfig
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    try:                                # the engine module loads liblrp_hip.so (a built artefact): without it this check has nothing to import
        import lxt_amd.engine_gemma3 as e
    except (ImportError, OSError, RuntimeError) as exc:     # _lib.LrpLibraryError is a RuntimeError
        pytest.skip(f"liblrp_hip.so is not built here: {exc}")
    cfg = Gemma3TextConfig(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                           head_dim=32, layer_types=["sliding_attention", "full_attention"],
                           rope_parameters={"sliding_attention": {"rope_type": "default", "rope_theta": 10000.0},
                                            "full_attention": {"rope_type": "linear", "factor": 8.0, "rope_theta": 1e6}})
    c5 = e.config_from_hf(cfg)
    inv, att = ROPE_INIT_FUNCTIONS["linear"](cfg, "cpu", layer_type="full_attention")
    assert torch.equal(inv.float(), c5["rope"]["full_attention"][0]) and att == 1.0

    class C4:                                                   # a transformers-4.x style config object
        pass
    c4 = C4()
    c4.__dict__.update(dict(model_type="gemma3_text", layer_types=["sliding_attention", "full_attention"], head_dim=32, rope_theta=1e6,
                            rope_local_base_freq=10000.0, rope_scaling={"rope_type": "linear", "factor": 8.0}, hidden_size=128, intermediate_size=256,
                            num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, vocab_size=512, rms_norm_eps=1e-6,
                            query_pre_attn_scalar=32, sliding_window=16, hidden_activation="gelu_pytorch_tanh"))
    c4c = e.config_from_hf(c4)
    for lt in ("sliding_attention", "full_attention"):
        assert torch.equal(c4c["rope"][lt][0], c5["rope"][lt][0])
    c4.rope_scaling = {"rope_type": "dynamic", "factor": 2.0}
    with pytest.raises(NotImplementedError):
        e.config_from_hf(c4)
