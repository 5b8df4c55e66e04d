"""ctypes binding of liblrp_hip.so (C ABI: include/lrp_hip.h).

The library is the product: if it is missing or does not export a symbol the header declares,
importing this module raises -- there is no CPU / PyTorch fallback anywhere in the package.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblrp_hip.so")
HEADER_PATH = os.path.abspath(os.path.join(_HERE, "..", "include", "lrp_hip.h"))

F32, BF16 = 0, 1
ACT = {"silu": 0, "gelu_tanh": 1, "gelu_pytorch_tanh": 1, "gelu": 2, "tanh": 3}
ERRORS = {-1: "LRP_EINVAL (bad argument)", -2: "LRP_EALIGN (pointer / leading dimension alignment)",
          -3: "LRP_ESHAPE (unsupported shape)", -4: "LRP_ELAUNCH (HIP launch failed)"}

_CTYPE = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float,
          "void*": ctypes.c_void_p, "float*": ctypes.c_void_p, "int*": ctypes.c_void_p,
          "char*": ctypes.c_char_p}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype_str, [argtype_str...])} for every function the header declares."""
    with open(path) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(lrp_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.replace("const", "").strip()
                base = a.rsplit(" ", 1)[0].strip() if " " in a else a
                if "*" in a:
                    base = a[: a.rindex("*") + 1].replace(" ", "")
                types.append(base)
        out[name] = (ret, types)
    return out


class LrpLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise LrpLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no fallback path.")
    # PyTorch wheels ship their OWN libamdhip64; the tensors this binding receives live in THAT runtime.  Load torch first so that the
    # dynamic linker binds liblrp_hip.so to the runtime already in the process -- with `import lxt_amd` ahead of `import torch` the library
    # would pull in the system ROCm's copy and every launch would fail with hipErrorNoDevice (two HIP runtimes in one process).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    decls = parse_header()
    for name, (ret, types) in decls.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LrpLibraryError(f"liblrp_hip.so does not export {name} declared in lrp_hip.h") from e
        fn.restype = {"int": ctypes.c_int, "int64_t": ctypes.c_int64}.get(ret, ctypes.c_char_p)
        fn.argtypes = [_CTYPE[t] for t in types]
    return lib, decls


lib, DECLS = _load()


def check(rc, name):
    if rc != 0:
        extra = f" hip error {lib.lrp_last_hip_error()}" if rc == -4 else ""
        raise RuntimeError(f"{name} failed: {ERRORS.get(rc, rc)}{extra}")
