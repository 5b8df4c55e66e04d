"""HIP-backed mirror of lxt.explicit: functional rules, rule modules, Composite (module swap)."""
from . import functional, rules, modules, special, check  # noqa: F401
from .core import Composite  # noqa: F401
