"""GPT-2: LayerNorm, Conv1D projections (c_attn / c_fc / c_proj on the HIP GEMM via patches.conv1d_forward; the lm_head is
an nn.Linear), plain MLP with the identity rule on its activation (ref wiring: lxt/efficient/models/gpt2.py:11-32)"""
from functools import partial

from torch.nn import LayerNorm
from transformers.models.gpt2 import modeling_gpt2 as MODELING_MODULE
from transformers.pytorch_utils import Conv1D

from ..patches import layer_norm_forward, conv1d_forward, patch_method
from ..rules import identity_rule_implicit
from ._maps import decoder_maps


def mlp_forward(self, hidden_states):
    return self.c_proj(identity_rule_implicit(self.act, self.c_fc(hidden_states)))


attnLRP, cp_LRP = decoder_maps(MODELING_MODULE, MODELING_MODULE.GPT2MLP, LayerNorm, norm_forward=layer_norm_forward,
                               mlp_forward=mlp_forward, cp_mlp_forward=mlp_forward, linear=True)
for _m in (attnLRP, cp_LRP):        # the modeling module (attention patch) stays last
    _mod_patch = _m.pop(MODELING_MODULE)
    _m[Conv1D] = partial(patch_method, conv1d_forward, keep_original=True)
    _m[MODELING_MODULE] = _mod_patch
