"""Per-family patch maps (ref: lxt/efficient/models/__init__.py:29-51).  A family whose HF module
cannot be imported in the running transformers version is skipped, like the reference does for bert
and vit_torch."""
import importlib
import warnings

DEFAULT_MAP = {}
_FAMILIES = ("llama", "qwen2", "qwen3", "gemma3", "gpt2", "bert", "vit_torch")

for _name in _FAMILIES:
    try:
        _m = importlib.import_module(f"{__name__}.{_name}")
        if _m.MODELING_MODULE is None:          # vit_torch without torchvision: usable with an explicit patch_map only
            continue
        DEFAULT_MAP[_m.MODELING_MODULE] = _m.attnLRP
    except Exception as _e:  # noqa: BLE001
        warnings.warn(f"lxt_amd.efficient.models.{_name} disabled: {_e}")


def get_default_map(module):
    if module in DEFAULT_MAP:
        return DEFAULT_MAP[module]
    supported = ", ".join(k.__name__ for k in DEFAULT_MAP)
    raise ValueError(f"{module.__name__} not yet supported. Supported models are: {supported} "
                     f"Please provide a custom patch_map.")
