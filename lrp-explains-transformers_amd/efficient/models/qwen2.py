"""Qwen2: Llama-style decoder with biased q/k/v projections (ref wiring: lxt/efficient/models/qwen2.py)"""
from transformers.models.qwen2 import modeling_qwen2 as MODELING_MODULE

from ._maps import decoder_maps

attnLRP, cp_LRP = decoder_maps(MODELING_MODULE, MODELING_MODULE.Qwen2MLP, MODELING_MODULE.Qwen2RMSNorm)
