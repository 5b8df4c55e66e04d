"""Composite: attach rules to the sub-modules of a model by type / name (module swap), apply canonizers first, and -- for models torch.fx can
trace -- re-target function calls in the traced graph to their LRP counterparts.

ref: lxt/explicit/core.py:25-152 (register / _parse_rules / _iterate_children / _attach_module_rule), :155-227 (_iterate_graph /
_attach_function_rule / _check_already_wrapped), :339-366 (remove / context).  Differences, all forced by the environment:
  * the reference traces with `transformers.utils.fx.HFTracer`, which upstream removed (SURVEY.md finding 9: the fx half is dead for HF models);
    here the default tracer is a plain `torch.fx.Tracer` that treats rule wrappers as leaves, so FUNCTION rules work for any module torch.fx can
    trace symbolically (custom torch models), and a HF model raises torch.fx's own tracing error instead of silently skipping its functions;
    the function-level rule sites of the supported HF families live in lxt_amd.explicit.models.* instead;
  * zennit is not installed: `zennit_composite` is refused."""
import inspect
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
        for c in self.canonizers:        # ref :35-37
            if isinstance(c, type):
                raise ValueError(f"You must call the canonizer {c}(). You passed the class instead of an instance.")
        self.canonizer_instances = []
        self.original_modules = []
        self.function_summary = {}

    # ------------------------------------------------------------------------------------------------------------------ register
    def register(self, parent, dummy_inputs=None, tracer=None, verbose=False, no_grad=True):
        """-> the model to call: `parent` itself (module rules only, modified in place) or the torch.fx GraphModule that carries the function
        rules (ref :42-92)."""
        if no_grad:
            for p in parent.parameters():
                p.requires_grad = False
        for canonizer in self.canonizers:            # ref :63-72: canonizers first, then the rules
            try:
                instances = canonizer.apply(parent, verbose)
            except TypeError:                        # zennit-style canonizers take no `verbose`
                instances = canonizer.apply(parent)
            self.canonizer_instances.extend(instances or [])
        module_map, fn_map = self._parse_rules(self.layer_map)
        if module_map:
            self._iterate_children(parent, module_map, verbose)
        if fn_map or dummy_inputs:
            parent = self._iterate_graph(parent, dummy_inputs, fn_map, module_map, tracer, verbose)
        return parent

    @staticmethod
    def _parse_rules(layer_map):
        """ref :94-106: module types and module NAMES are module rules, any other callable is a function rule"""
        module_map, fn_map = {}, {}
        for key, value in layer_map.items():
            if isinstance(key, str) or (isinstance(key, type) and issubclass(key, nn.Module)):
                module_map[key] = value
            elif callable(key):
                fn_map[key] = value
            else:
                raise ValueError(f"Key {key} must be a subclass of nn.Module, a string or a callable function.")
        return module_map, fn_map

    def _iterate_children(self, parent, module_map, verbose=False):
        for name, child in list(parent.named_children()):
            rule = self._find_rule(name, child, module_map)
            if rule is not None:
                new = self._attach_module_rule(child, rule)
                setattr(parent, name, new)
                self.original_modules.append((parent, name, child))
                if verbose:
                    print(f"{name}: {type(child).__name__} -> {rule.__name__}")
                if not isinstance(new, WrapModule):          # a replaced module (lm.* classes) may have children of its own (ref :136-138)
                    self._iterate_children(new, module_map, verbose)
            else:
                self._iterate_children(child, module_map, verbose)

    @staticmethod
    def _find_rule(name, child, module_map):
        if isinstance(child, WrapModule):
            return None
        for key, rule in module_map.items():
            if (isinstance(key, str) and key == name) or (isinstance(key, type) and isinstance(child, key)):
                return rule
        return None

    @staticmethod
    def _attach_module_rule(child, rule):
        if isinstance(rule, type) and issubclass(rule, WrapModule):
            return rule(child)
        if rule in INIT_MODULE_MAPPING:
            return INIT_MODULE_MAPPING[rule](child, rule)
        raise ValueError(f"Rule {rule} must be a subclass of WrapModule or one of {[r.__name__ for r in INIT_MODULE_MAPPING]}")

    # ------------------------------------------------------------------------------------------------------------------ function rules
    def _iterate_graph(self, model, dummy_inputs, fn_map, module_map, tracer=None, verbose=False):
        """ref :155-176: trace, re-target `call_function` nodes whose target is a key of fn_map (never inside a module that already carries a
        rule), recompile.  `dummy_inputs` names the forward arguments that stay symbolic; every other argument is fixed to its default."""
        import torch.fx as fx
        if not isinstance(dummy_inputs, dict) or not dummy_inputs:
            raise ValueError("function rules need dummy_inputs: a dict {forward argument name: example tensor} (ref lxt/explicit/core.py:158-160)")
        rule_types = tuple(t for t in (set(module_map.values()) | set(INIT_MODULE_MAPPING)) if isinstance(t, type))

        class _Tracer(fx.Tracer):
            def is_leaf_module(self, m, qualname):           # rule wrappers hold autograd Functions: opaque to the tracer
                return isinstance(m, (WrapModule,) + rule_types) or super().is_leaf_module(m, qualname)

        sig = inspect.signature(model.forward)
        concrete = {n: p.default for n, p in sig.parameters.items()
                    if n not in dummy_inputs and p.default is not inspect.Parameter.empty and p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)}
        tr = (tracer or _Tracer)()
        graph = tr.trace(model, concrete_args=concrete or None)
        for node in graph.nodes:
            self._attach_function_rule(node, fn_map, rule_types)
        graph.lint()
        traced = fx.GraphModule(tr.root if hasattr(tr, "root") else model, graph)
        traced.recompile()
        if verbose:
            self.print_summary()
        return traced

    def _attach_function_rule(self, node, fn_map, rule_types):
        """ref :179-229"""
        if self._check_already_wrapped(node, rule_types):
            return False
        if node.op == "call_function":
            where = self._module_of(node)
            if node.target in fn_map:
                self.function_summary.setdefault(where, {})[node.target] = "replaced"
                node.target = fn_map[node.target]
                # torch.nn.functional wrappers reach the graph through __torch_function__ with their PRIVATE keywords spelled out
                # (F.softmax: _stacklevel); the rule function is called with the keywords it declares
                try:
                    params = inspect.signature(node.target).parameters
                    if not any(p_.kind == p_.VAR_KEYWORD for p_ in params.values()):
                        node.kwargs = {k: v for k, v in node.kwargs.items() if k in params}
                except (TypeError, ValueError):
                    pass
                return True
            self.function_summary.setdefault(where, {}).setdefault(node.target, "not replaced")
        elif node.op == "call_method":      # as the reference: methods (tensor.add, ...) are reported, never replaced
            self.function_summary.setdefault(self._module_of(node), {}).setdefault(node.target, "method: not replaced")
        return False

    @staticmethod
    def _stack_types(node):
        out = []
        for v in (node.meta.get("nn_module_stack") or {}).values():
            out.append(v[1] if isinstance(v, tuple) else v)          # torch >= 2.0 records (qualified name, type)
        return out

    def _check_already_wrapped(self, node, rule_types):
        """ref :232-251: a function inside a module that already carries a rule is left alone"""
        return any(isinstance(t, type) and issubclass(t, rule_types + (WrapModule,)) for t in self._stack_types(node))

    def _module_of(self, node):
        ts = self._stack_types(node)
        return getattr(ts[-1], "__name__", str(ts[-1])) if ts else "Root"

    def print_summary(self):
        """ref :300-333 (plain text: tabulate is a formatting dependency only)"""
        for module, functions in self.function_summary.items():
            for fn, rating in functions.items():
                print(f"{module:32s} {getattr(fn, '__name__', str(fn)):32s} {rating}")

    # ------------------------------------------------------------------------------------------------------------------ remove / context
    def remove(self):
        """ref :339-362: module and canonizer replacements are reverted; a traced GraphModule (function rules) is simply dropped by the caller"""
        for parent, name, child in self.original_modules:
            setattr(parent, name, child)
        for instance in self.canonizer_instances:
            instance.remove()
        self.original_modules = []
        self.canonizer_instances = []

    @contextmanager
    def context(self, module, **kwargs):
        try:
            yield self.register(module, **kwargs)
        finally:
            self.remove()
