"""ctypes binding of libkd_hip.so.  Prototypes are parsed from include/kd_hip.h so the Python
side can never drift from the C ABI (tests/test_abi.py checks every declared symbol is exported)."""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
_ROOT = os.path.dirname(_PKG)
# KD_HIP_LIB: a development build of the same library (instrumented kernels under tools/dbg/); never a different backend
SO_PATH = os.environ.get("KD_HIP_LIB") or os.path.join(_PKG, "csrc", "libkd_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "kd_hip.h")


class KDError(RuntimeError):
    pass


_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t,
    "uint64_t": ctypes.c_uint64,
    "void": None,
}


def _ctype(decl: str):
    decl = decl.replace("const", "").strip()
    if "*" in decl:
        return ctypes.c_char_p if decl.replace(" ", "") == "char*" else ctypes.c_void_p
    return _CT[decl]


def parse_header(path: str = HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every `kd_*` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"((?:const\s+)?\w+\s*\*?)\s*\b(kd_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                ty = a.rsplit(None, 1)[0] if not a.endswith("*") else a
                if "*" in a:
                    ty = a[: a.rindex("*") + 1]
                argtypes.append(_ctype(ty))
        protos[name] = (_ctype(ret), argtypes)
    return protos


class _Lib:
    def __init__(self):
        if not os.path.exists(SO_PATH):
            raise KDError(
                f"HIP library not built: {SO_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C csrc`). There is no CPU fallback for the product path.")
        self._dll = ctypes.CDLL(SO_PATH)
        self.protos = parse_header()
        for name, (res, args) in self.protos.items():
            fn = getattr(self._dll, name)          # AttributeError here == header/library mismatch
            fn.restype = res
            fn.argtypes = args
        self._dll.kd_last_error_string.restype = ctypes.c_char_p

    def __getattr__(self, name):
        return getattr(self._dll, name)

    def call(self, name: str, *args):
        """Call an int-returning entry point and raise KDError on a non-zero status."""
        rc = getattr(self._dll, name)(*args)
        if rc != 0:
            msg = self._dll.kd_last_error_string()
            raise KDError(f"{name} failed with status {rc}: {msg.decode() if msg else ''}")


lib = _Lib()


def require_gpu_tensor(t, what: str):
    if not t.is_cuda:
        raise KDError(f"{what}: expected a tensor on an MI355X device, got device={t.device}. "
                      "The product path runs only through the HIP library (no CPU fallback).")
