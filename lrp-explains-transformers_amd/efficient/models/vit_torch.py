"""torchvision ViT: AttnLRP outside the attention, CP-LRP inside it (ref: lxt/efficient/models/vit_torch.py:7-11).
The map's keys are torch.nn classes, so it applies to any model built from them (torchvision is only needed for
MODELING_MODULE, i.e. for monkey_patch(vision_transformer) without an explicit map).  nn.Linear and the patch-embedding
Conv2d are routed to the HIP GEMM as in the other maps; the reference pairs this map with zennit's Gamma rule for those
two layer types (docs/source/quickstart.rst:376-384) -- zennit is not available here; the native counterpart is
lxt_amd.efficient.gamma.GammaComposite (parity unpinned, see that file).
The torch.nn patches only act on instances of an explained model (patches.adopt): with torchvision present the
VisionTransformer class adopts its instances on the first call; a model built from plain torch.nn modules is handed to
`lxt_amd.efficient.adopt(model)` once by the user."""
from functools import partial

from torch import nn

from ..patches import (patch_method, non_linear_forward, layer_norm_forward, cp_multi_head_attention_forward, linear_forward,
                       conv2d_patch_forward, dropout_forward, _adopting_call)

try:
    from torchvision.models import vision_transformer as MODELING_MODULE
except Exception:  # noqa: BLE001  (torchvision absent: the map is still usable with an explicit patch_map)
    MODELING_MODULE = None

cp_LRP = {
    nn.GELU: partial(patch_method, non_linear_forward, keep_original=True),
    nn.LayerNorm: partial(patch_method, layer_norm_forward, keep_original=True),
    nn.MultiheadAttention: partial(patch_method, cp_multi_head_attention_forward, keep_original=True),
    nn.Dropout: partial(patch_method, dropout_forward),
    nn.Linear: partial(patch_method, linear_forward, keep_original=True),
    nn.Conv2d: partial(patch_method, conv2d_patch_forward, keep_original=True),
}
if MODELING_MODULE is not None:
    def _adopt_vit(cls):
        cls.__call__ = _adopting_call
        return True
    cp_LRP[MODELING_MODULE.VisionTransformer] = _adopt_vit
attnLRP = cp_LRP
