"""ref: lxt/efficient/models/gemma3.py:11-26 (the reference patches Gemma3RMSNorm._norm; here the whole
(1+w) RMSNorm forward is one fused kernel)"""
from functools import partial

from torch.nn import Conv2d, Dropout, Linear
from transformers.models.gemma3 import modeling_gemma3
from transformers.models.gemma3.modeling_gemma3 import Gemma3MLP, Gemma3RMSNorm

from ..patches import (patch_method, patch_attention, patch_cp_attention, gemma3_rms_norm_forward, gated_mlp_forward,
                       cp_gated_mlp_forward, dropout_forward, linear_forward, conv2d_patch_forward)

MODELING_MODULE = modeling_gemma3

attnLRP = {
    Gemma3MLP: partial(patch_method, gated_mlp_forward),
    Gemma3RMSNorm: partial(patch_method, gemma3_rms_norm_forward),
    Dropout: partial(patch_method, dropout_forward),
    Linear: partial(patch_method, linear_forward),
    Conv2d: partial(patch_method, conv2d_patch_forward),      # SigLIP patch embedding (image branch) as a GEMM
    modeling_gemma3: patch_attention,
}

cp_LRP = {
    Gemma3MLP: partial(patch_method, cp_gated_mlp_forward),
    Gemma3RMSNorm: partial(patch_method, gemma3_rms_norm_forward),
    Dropout: partial(patch_method, dropout_forward),
    Linear: partial(patch_method, linear_forward),
    Conv2d: partial(patch_method, conv2d_patch_forward),
    modeling_gemma3: patch_cp_attention,
}
