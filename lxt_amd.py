"""Import alias: the package directory is ``lrp-explains-transformers_amd`` (not a legal Python
identifier), so ``import lxt_amd`` loads that directory as the package ``lxt_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lrp-explains-transformers_amd")
_spec = importlib.util.spec_from_file_location("lxt_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lxt_amd"] = _mod
_spec.loader.exec_module(_mod)
