"""ref: lxt/efficient/models/gpt2.py:11-32 (Conv1D MLP with the identity rule on the activation)"""
from functools import partial

from torch.nn import Dropout, LayerNorm
from transformers.models.gpt2 import modeling_gpt2
from transformers.models.gpt2.modeling_gpt2 import GPT2MLP

from ..patches import patch_method, patch_attention, patch_cp_attention, layer_norm_forward, dropout_forward
from ..rules import identity_rule_implicit

MODELING_MODULE = modeling_gpt2


def mlp_forward(self, hidden_states):
    hidden_states = self.c_fc(hidden_states)
    hidden_states = identity_rule_implicit(self.act, hidden_states)
    return self.c_proj(hidden_states)


attnLRP = {
    GPT2MLP: partial(patch_method, mlp_forward),
    LayerNorm: partial(patch_method, layer_norm_forward),
    Dropout: partial(patch_method, dropout_forward),
    modeling_gpt2: patch_attention,
}

cp_LRP = {
    GPT2MLP: partial(patch_method, mlp_forward),
    LayerNorm: partial(patch_method, layer_norm_forward),
    Dropout: partial(patch_method, dropout_forward),
    modeling_gpt2: patch_cp_attention,
}
