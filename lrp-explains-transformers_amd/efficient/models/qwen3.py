"""Qwen3: Llama-style decoder with q/k RMSNorm inside the attention block (ref wiring: lxt/efficient/models/qwen3.py)"""
from transformers.models.qwen3 import modeling_qwen3 as MODELING_MODULE

from ._maps import decoder_maps

attnLRP, cp_LRP = decoder_maps(MODELING_MODULE, MODELING_MODULE.Qwen3MLP, MODELING_MODULE.Qwen3RMSNorm)
