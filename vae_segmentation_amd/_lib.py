"""ctypes binding of libvaeseg.so — the only way the package reaches the GPU kernels.

The prototypes are read from ``include/vaeseg.h`` (the single source of truth for the C ABI), so a
declaration that the shared library does not export fails at import time, loudly.  There is no CPU
or pure-PyTorch fallback anywhere in this package: if the library is missing, importing the ops raises.
"""
import ctypes
import os
import re

# PyTorch-ROCm ships its own libamdhip64.so; libvaeseg.so links against the same SONAME.  torch must be loaded FIRST so
# that the single HIP runtime in the process is torch's (the one that owns the device context and the streams we are
# handed): loading libvaeseg.so first pulls /opt/rocm's copy instead and torch then fails with hipErrorNoDevice (100).
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "vaeseg.h")
LIB_PATH = os.environ.get("VS_LIBVAESEG") or os.path.join(_HERE, "libvaeseg.so")      # VS_LIBVAESEG: an alternative build (A/B measurements)

VS_F32, VS_BF16, VS_F16 = 0, 1, 2
VS_CONV_K3, VS_CONV_K2S2, VS_CONV_T2S2 = 0, 1, 2
VS_PACK_ROWS_D0, VS_PACK_ROWS_D1_FLIP, VS_PACK_SCATTER_D1 = 0, 1, 2

_SCALARS = {"int": ctypes.c_int, "unsigned long long": ctypes.c_ulonglong, "long long": ctypes.c_longlong, "float": ctypes.c_float,
            "size_t": ctypes.c_size_t, "double": ctypes.c_double}


def _ctype(decl):
    decl = decl.strip()
    if "*" in decl:
        return ctypes.c_void_p
    decl = re.sub(r"\bconst\b", "", decl).strip()
    # drop the parameter name
    for key in sorted(_SCALARS, key=len, reverse=True):
        if decl == key or decl.startswith(key + " "):
            return _SCALARS[key]
    raise ValueError("unparsed C type: %r" % decl)


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every ``vs_*`` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(vs_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.endswith("*"):
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _ctype(ret)
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        protos[name] = (restype, argtypes)
    return protos


class VaesegError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("libvaeseg.so not found at %s — build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` or `make -C vae_segmentation_amd/csrc`; there is no fallback path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(lib, name)          # AttributeError if the header declares what the .so lacks
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()


def check(code, what=""):
    if code != 0:
        msg = lib.vs_strerror(code)
        raise VaesegError("libvaeseg %s failed (%d): %s" % (what, code, msg.decode() if msg else "?"))
