"""MI355X-native AttnLRP engine (package directory ``lrp-explains-transformers_amd``; import it
as ``lxt_amd`` through the one-file alias at the repo root).

Layout
------
csrc/        hand-written HIP kernels for gfx950 + the C ABI (include/lrp_hip.h) -> liblrp_hip.so
_lib.py      ctypes binding (fails loudly if the library is missing -- no fallback)
ops.py       tensor-level wrappers: torch tensors in, raw pointers + current HIP stream out
engine.py    whole-model explain(): fused forward + LRP backward for Llama-style decoders
efficient/   mirror of lxt.efficient  (monkey_patch, rules, patches, model maps)
explicit/    mirror of lxt.explicit   (functional rules, rule modules, Composite)
dist.py      one-process-per-GPU sharding of explanation jobs over RCCL
"""
from . import _lib  # noqa: F401  (raises if liblrp_hip.so is absent)

__version__ = "0.1.0"
