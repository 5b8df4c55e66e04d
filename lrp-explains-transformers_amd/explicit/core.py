"""Composite: attach rules to the sub-modules of a model by type (module swap).
ref: lxt/explicit/core.py:25-152,339-366.  The torch.fx function-rewrite half of the reference
(core.py:155-227) depends on transformers.utils.fx, removed upstream, and is out of scope
(SURVEY.md 2.1 #13): passing function keys in layer_map raises."""
from contextlib import contextmanager

import torch.nn as nn

from .rules import WrapModule
from .modules import INIT_MODULE_MAPPING


class Composite:
    def __init__(self, layer_map, canonizers=None, zennit_composite=None):
        if zennit_composite is not None:
            raise NotImplementedError("zennit composites are not supported (zennit is not installed here)")
        self.layer_map = dict(layer_map)
        self.canonizers = list(canonizers or [])
        self.original_modules = []

    def register(self, parent, dummy_inputs=None, tracer=None, verbose=False, no_grad=True):
        for k in self.layer_map:
            if not (isinstance(k, type) and issubclass(k, nn.Module)):
                raise NotImplementedError("function rules need torch.fx tracing of HF models, which is out of scope")
        if no_grad:
            for p in parent.parameters():
                p.requires_grad = False
        self._iterate_children(parent, verbose)
        return parent

    def _iterate_children(self, parent, verbose):
        for name, child in list(parent.named_children()):
            rule = self._find_rule(child)
            if rule is not None:
                new = self._attach_module_rule(child, rule)
                setattr(parent, name, new)
                self.original_modules.append((parent, name, child))
                if verbose:
                    print(f"{name}: {type(child).__name__} -> {rule.__name__}")
            else:
                self._iterate_children(child, verbose)

    def _find_rule(self, child):
        for typ, rule in self.layer_map.items():
            if isinstance(child, typ) and not isinstance(child, WrapModule):
                return rule
        return None

    @staticmethod
    def _attach_module_rule(child, rule):
        if isinstance(rule, type) and issubclass(rule, WrapModule):
            return rule(child)
        if rule in INIT_MODULE_MAPPING:
            return INIT_MODULE_MAPPING[rule](child, rule)
        raise ValueError(f"no initialiser for rule {rule}")

    def remove(self):
        for parent, name, child in self.original_modules:
            setattr(parent, name, child)
        self.original_modules = []

    @contextmanager
    def context(self, module, **kwargs):
        try:
            yield self.register(module, **kwargs)
        finally:
            self.remove()
