"""monkey_patch -- same signature and failure behaviour as the reference
(ref: lxt/efficient/core.py:20-43)."""
from warnings import warn

from .models import get_default_map


def monkey_patch(module, patch_map=None, verbose=False):
    """Patch the classes/functions of a HuggingFace modeling module so that a plain
    ``logit.backward()`` propagates AttnLRP relevance through the HIP kernels.

    patch_map: {class_or_module: callable(target) -> bool}; None selects the default map of
    `module` (ValueError if the module is not supported).  A patch returning False produces a
    warning and the loop continues; nothing is ever un-patched."""
    if patch_map is None:
        patch_map = get_default_map(module)
    for target, patch in patch_map.items():
        success = patch(target)
        if not success:
            warn(f"Failed to patch {target.__name__}. Skipping...")
        elif verbose:
            print(f"Patched {target.__name__}")
