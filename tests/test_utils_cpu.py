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
