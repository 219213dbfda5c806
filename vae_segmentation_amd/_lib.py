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
VS_CONV_K3, VS_CONV_K2S2, VS_CONV_T2S2, VS_CONV_UP = 0, 1, 2, 3
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


DET_LIB_PATH = os.path.join(_HERE, "libvaeseg_det.so")


def _load(path):
    if not os.path.exists(path):
        raise ImportError("%s not found — build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` or `make -C vae_segmentation_amd/csrc`; there is no fallback path" % path)
    cdll = ctypes.CDLL(path)
    for name, (restype, argtypes) in parse_header().items():
        if os.environ.get("VS_LIBVAESEG") and not hasattr(cdll, name):
            continue                     # an older build loaded for a same-box A/B measurement: entry points it lacks fail when called
        fn = getattr(cdll, name)         # AttributeError if the header declares what the .so lacks
        fn.restype = restype
        fn.argtypes = argtypes
    return cdll


class _Lib:
    """The library the package calls into: libvaeseg.so (throughput build, fp64-atomic statistics) or — after use_deterministic(True) /
    under env VS_DETERMINISTIC=1 — libvaeseg_det.so, the same sources compiled with -DVS_DET_BUILD=1 (statistics summed with commuting integer
    atomics: bit-reproducible parity runs, csrc/common.h).  Both export every symbol of include/vaeseg.h; attribute access goes to the active one."""

    def __init__(self):
        self._fast = _load(LIB_PATH)
        self._det = None
        self._active = self._fast

    def use_deterministic(self, on):
        if on:
            if self._det is None:
                self._det = _load(DET_LIB_PATH)
                if not self._det.vs_get_deterministic():
                    raise ImportError("%s is not a deterministic build (VS_DET_BUILD)" % DET_LIB_PATH)
                # the second build starts from the first one's CURRENT tuning switches (ops.set_config may have moved them), not from the environment again
                buf = ctypes.create_string_buffer(self._fast.vs_config_bytes())
                if self._fast.vs_get_config(ctypes.addressof(buf)) == 0:
                    self._det.vs_set_config(ctypes.addressof(buf))
            self._active = self._det
        else:
            self._active = self._fast

    def loaded(self):
        """every build loaded so far (a tuning configuration is set on all of them: ops.set_config)"""
        return [l for l in (self._fast, self._det) if l is not None]

    def __getattr__(self, name):
        return getattr(self._active, name)


lib = _Lib()
if os.environ.get("VS_DETERMINISTIC", "0") == "1":
    lib.use_deterministic(True)


def check(code, what=""):
    if code != 0:
        msg = lib.vs_strerror(code)
        raise VaesegError("libvaeseg %s failed (%d): %s" % (what, code, msg.decode() if msg else "?"))
