"""GPT-2: LayerNorm, Conv1D projections (left on their own forward), plain MLP with the identity rule on its activation
(ref wiring: lxt/efficient/models/gpt2.py:11-32)"""
from torch.nn import LayerNorm
from transformers.models.gpt2 import modeling_gpt2 as MODELING_MODULE

from ..patches import layer_norm_forward
from ..rules import identity_rule_implicit
from ._maps import decoder_maps


def mlp_forward(self, hidden_states):
    return self.c_proj(identity_rule_implicit(self.act, self.c_fc(hidden_states)))


attnLRP, cp_LRP = decoder_maps(MODELING_MODULE, MODELING_MODULE.GPT2MLP, LayerNorm, norm_forward=layer_norm_forward,
                               mlp_forward=mlp_forward, cp_mlp_forward=mlp_forward, linear=False)
