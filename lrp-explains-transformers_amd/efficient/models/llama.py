"""Llama: gated-SiLU MLP, RMSNorm (ref wiring: lxt/efficient/models/llama.py:9-21)"""
from transformers.models.llama import modeling_llama as MODELING_MODULE

from ._maps import decoder_maps

attnLRP, cp_LRP = decoder_maps(MODELING_MODULE, MODELING_MODULE.LlamaMLP, MODELING_MODULE.LlamaRMSNorm)

# the whole decoder layer as one fused autograd node where it applies (bf16, M = B S rows, plain causal batches; patches.decoder_layer_forward
# hands everything else to HF's own forward over the per-module patches): the engine's K1n launch sequence behind the drop-in surface
from functools import partial  # noqa: E402

from .. import patches as _P  # noqa: E402

for _tbl in (attnLRP,):
    _mod_patch = _tbl.pop(MODELING_MODULE)
    _tbl[MODELING_MODULE.LlamaDecoderLayer] = partial(_P.patch_method, _P.decoder_layer_forward, keep_original=True)
    _tbl[MODELING_MODULE] = _mod_patch               # the modeling module stays last (ref lxt/efficient/models/__init__.py)
