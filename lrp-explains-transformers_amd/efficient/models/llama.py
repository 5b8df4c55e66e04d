"""Llama: gated-SiLU MLP, RMSNorm (ref wiring: lxt/efficient/models/llama.py:9-21)"""
from transformers.models.llama import modeling_llama as MODELING_MODULE

from ._maps import decoder_maps

attnLRP, cp_LRP = decoder_maps(MODELING_MODULE, MODELING_MODULE.LlamaMLP, MODELING_MODULE.LlamaRMSNorm)
