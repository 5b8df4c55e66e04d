"""Presentation helpers of the reference (ref: lxt/utils.py:68-117): token clean-up and heat-map rendering of per-token
relevance.  Host-side text processing only -- no kernels.  Differences from the reference: the diverging colour maps
are computed here (no matplotlib import), the .tex source is kept when no LaTeX backend is installed (the reference
fails inside subprocess), and an HTML renderer is offered for machines without LaTeX."""
import html
import os
import shutil
import subprocess
from pathlib import Path

_LATEX_SPECIAL = ("\\", "&", "%", "$", "#", "_", "{", "}")


def clean_tokens(words):
    """word-piece / sentence-piece / byte-level BPE tokens -> printable words with their leading blanks; LaTeX specials
    escaped (ref: utils.py:95-117).  Raises ValueError when the tokenisation scheme is not recognised."""
    words = list(words)
    if any("\u2581" in w for w in words):                     # sentence-piece
        words = [w.replace("\u2581", " ") for w in words]
    elif any("\u0120" in w for w in words):                   # byte-level BPE
        words = [w.replace("\u0120", " ") for w in words]
    elif any("##" in w for w in words):                       # word-piece
        words = [w.replace("##", "") if "##" in w else " " + w for w in words]
        words[0] = words[0].strip()
    else:
        raise ValueError("The tokenization scheme is not recognized.")
    out = []
    for w in words:
        for ch in _LATEX_SPECIAL:
            if ch in w:
                w = w.replace(ch, "\\" + ch)
        out.append(w)
    return out


def _colour(value, cmap="bwr"):
    """diverging map on [-1, 1] -> (r, g, b) in 0..255; 'bwr' blue-white-red, 'seismic' with darker ends"""
    v = max(-1.0, min(1.0, float(value)))
    if cmap not in ("bwr", "seismic"):
        raise ValueError(f"colour map {cmap!r} is not available (bwr, seismic)")
    a = abs(v)
    lo = 1.0 - a                                              # white at 0, saturated at +-1
    r, g, b = (1.0, lo, lo) if v >= 0 else (lo, lo, 1.0)
    if cmap == "seismic" and a > 0.5:                         # darken the outer half
        k = 1.0 - (a - 0.5)
        r, g, b = r * k if v < 0 else r * (0.5 + 0.5 * k), g * k, b * k if v >= 0 else b * (0.5 + 0.5 * k)
    return int(round(r * 255)), int(round(g * 255)), int(round(b * 255))


def _generate_latex(words, relevances, cmap="bwr"):
    body = []
    for w, rel in zip(words, relevances):
        r, g, b = _colour(rel, cmap)
        body.append(("" if not w.startswith(" ") else " ") + f"\\colorbox[RGB]{{{r},{g},{b}}}{{\\strut {w}}}")
    return ("\\documentclass[varwidth=200mm]{standalone}\n\\usepackage[dvipsnames]{xcolor}\n\\begin{document}\n\\fbox{\n"
            "\\parbox{\\textwidth}{\n\\setlength\\fboxsep{0pt}\n" + "".join(body) + "}}\n\\end{document}\n")


def _check(words, relevances):
    rel = [float(r) for r in relevances]
    if len(words) != len(rel):
        raise AssertionError("The number of words and relevances must be the same.")
    if rel and (min(rel) < -1 or max(rel) > 1):
        raise AssertionError("The relevances must be normalized between -1 and 1.")
    return rel


def pdf_heatmap(words, relevances, cmap="bwr", path="heatmap.pdf", delete_aux_files=True, backend="xelatex"):
    """colour every word by its relevance (normalised to [-1, 1]) and typeset the sentence (ref: utils.py:68-92).
    Returns the path of the PDF, or of the kept .tex source when `backend` is not installed."""
    rel = _check(words, relevances)
    path = Path(path)
    os.makedirs(path.parent if str(path.parent) else ".", exist_ok=True)
    tex = path.with_suffix(".tex")
    tex.write_text(_generate_latex(words, rel, cmap))
    if backend not in ("xelatex", "pdflatex"):
        raise ValueError("backend must be 'xelatex' or 'pdflatex'")
    if shutil.which(backend) is None:
        return str(tex)
    subprocess.call([backend, "--output-directory", str(path.parent), str(tex)])
    if delete_aux_files:
        for suffix in (".aux", ".log", ".tex"):
            try:
                os.remove(path.with_suffix(suffix))
            except FileNotFoundError:
                pass
    return str(path)


def html_heatmap(words, relevances, cmap="bwr", path=None):
    """same picture as an HTML fragment (string; also written to `path` if given); LaTeX escapes are undone"""
    rel = _check(words, relevances)
    spans = []
    for w, r_ in zip(words, rel):
        for ch in _LATEX_SPECIAL:
            w = w.replace("\\" + ch, ch)
        r, g, b = _colour(r_, cmap)
        spans.append(f'<span style="background-color: rgb({r},{g},{b})">{html.escape(w)}</span>')
    doc = '<div style="font-family: monospace; white-space: pre-wrap">' + "".join(spans) + "</div>"
    if path is not None:
        Path(path).write_text(doc)
    return doc
