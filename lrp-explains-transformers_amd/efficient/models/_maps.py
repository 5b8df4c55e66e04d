"""One builder for the per-family patch tables.  Every decoder family wires the same four things -- its gated MLP class, its
norm class, nn.Dropout (+ nn.Linear and, where an image tower exists, the patch-embedding nn.Conv2d, which the reference
leaves on ATen) and the attention functions of its modeling module -- and differs only in the class names and in the
MLP / attention variant (AttnLRP vs CP-LRP).  The returned dicts keep the reference's contract (ref:
lxt/efficient/models/__init__.py:29-51, lxt/efficient/core.py:20-45): {class or module: callable(target) -> bool}, applied
in insertion order, the modeling module last."""
from functools import partial

from torch import nn

from .. import patches as P


def _table(first, shared, module, attention_patch):
    table = dict([first])
    table.update(shared)
    table[module] = attention_patch
    return table


def decoder_maps(module, mlp_cls, norm_cls, norm_forward=P.rms_norm_forward, mlp_forward=P.gated_mlp_forward,
                 cp_mlp_forward=P.cp_gated_mlp_forward, linear=True, conv_patch_embedding=False):
    """-> (attnLRP, cp_LRP) for one HuggingFace modeling module"""
    # torch.nn classes keep their original forward (keep_original): the patched forwards hand every instance that is not
    # part of an explained model back to it (patches.adopt / patches.linear_forward)
    shared = [(norm_cls, partial(P.patch_method, norm_forward, keep_original=norm_cls is nn.LayerNorm)),
              (nn.Dropout, partial(P.patch_method, P.dropout_forward))]
    if linear:
        shared.append((nn.Linear, partial(P.patch_method, P.linear_forward, keep_original=True)))
    if conv_patch_embedding:
        shared.append((nn.Conv2d, partial(P.patch_method, P.conv2d_patch_forward, keep_original=True)))
    attn = _table((mlp_cls, partial(P.patch_method, mlp_forward)), shared, module, P.patch_attention)
    cp = _table((mlp_cls, partial(P.patch_method, cp_mlp_forward)), shared, module, P.patch_cp_attention)
    return attn, cp
