"""HIP-backed mirror of lxt.efficient: monkey_patch + the three gradient-modifier primitives."""
from .core import monkey_patch  # noqa: F401
from .rules import identity_rule_implicit, divide_gradient, stop_gradient  # noqa: F401
