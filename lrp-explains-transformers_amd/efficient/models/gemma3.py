"""Gemma-3: (1 + w) RMSNorm as one fused kernel (the reference patches Gemma3RMSNorm._norm only: lxt/efficient/models/
gemma3.py:11-26), gelu-tanh gated MLP, sliding + global attention layers; with the image branch the SigLIP patch-embedding
Conv2d (stride = kernel) runs as a GEMM on the unfolded patches."""
from transformers.models.gemma3 import modeling_gemma3 as MODELING_MODULE

from ..patches import gemma3_rms_norm_forward
from ._maps import decoder_maps

attnLRP, cp_LRP = decoder_maps(MODELING_MODULE, MODELING_MODULE.Gemma3MLP, MODELING_MODULE.Gemma3RMSNorm,
                               norm_forward=gemma3_rms_norm_forward, conv_patch_embedding=True)
