"""BERT on HF's own modeling_bert (the reference vendors a patched copy, lxt/efficient/models/bert.py,
whose marked edits are :339,380,476-488 attention factors, :581 intermediate GELU, :790,806 pooler /
head activations; that copy does not run under transformers 5.x -- SURVEY.md finding 9).  The same
rule placement is obtained with a patch map."""
from functools import partial

from torch.nn import Dropout, LayerNorm, Linear
from transformers.models.bert import modeling_bert
from transformers.models.bert.modeling_bert import BertIntermediate, BertPooler

from ..patches import (patch_method, patch_attention, patch_cp_attention, layer_norm_forward, dropout_forward,
                       linear_forward)
from ..rules import identity_rule_implicit

MODELING_MODULE = modeling_bert


def intermediate_forward(self, hidden_states):
    return identity_rule_implicit(self.intermediate_act_fn, self.dense(hidden_states))


def pooler_forward(self, hidden_states):
    return identity_rule_implicit(self.activation, self.dense(hidden_states[:, 0]))


attnLRP = {
    BertIntermediate: partial(patch_method, intermediate_forward),
    BertPooler: partial(patch_method, pooler_forward),
    LayerNorm: partial(patch_method, layer_norm_forward, keep_original=True),
    Dropout: partial(patch_method, dropout_forward),
    Linear: partial(patch_method, linear_forward, keep_original=True),
    modeling_bert: patch_attention,
}

cp_LRP = dict(attnLRP)
cp_LRP[modeling_bert] = patch_cp_attention
