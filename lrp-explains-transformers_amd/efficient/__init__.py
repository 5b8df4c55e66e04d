"""HIP-backed mirror of lxt.efficient: monkey_patch + the three gradient-modifier primitives."""
from .core import monkey_patch  # noqa: F401
from .rules import identity_rule_implicit, divide_gradient, stop_gradient  # noqa: F401
from .patches import adopt  # noqa: F401


def monkey_patch_zennit(verbose=False):
    """ref: lxt/efficient/zennit_patches.py:65-79 re-wires zennit's BasicHook to the gradient x input framework so that zennit
    rules (Gamma ...) can be mixed with the efficient patches.  zennit is a third-party package the reference neither pins nor
    vendors and that is not installed here; the one rule the reference's ViT recipe takes from it is provided natively:
    `lxt_amd.efficient.gamma.GammaComposite([(nn.Conv2d, g1), (nn.Linear, g2)]).register(model)`.  Fails loudly."""
    raise NotImplementedError("lxt_amd does not patch zennit hooks; use lxt_amd.efficient.gamma.GammaComposite "
                              "(Gamma rule for nn.Linear / patch-embedding nn.Conv2d on the HIP GEMM) instead")
