"""ctypes binding of libpulpo_hip.so.  The prototypes are read from include/pulpo_hip.h, the single source of truth.

There is NO CPU fallback: if the shared library is missing this module raises at first use, and every operator in
pulpo_amd.ops refuses non-CUDA tensors.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

import torch  # noqa: F401  -- must be imported before the library so that torch's bundled HIP runtime is the one mapped

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "pulpo_hip.h")
# PULPO_HIP_LIB: load a differently built copy of the library (diagnostic ablation builds of scripts/ablate.py); default = the in-tree build
LIB_PATH = os.environ.get("PULPO_HIP_LIB") or os.path.join(_HERE, "csrc", "libpulpo_hip.so")

_CTYPE = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "size_t": ctypes.c_size_t,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "void": None,
}


def _arg_type(decl: str):
    decl = decl.strip()
    if decl.count("*") >= 2:                       # const float* const* / float* const*
        return ctypes.POINTER(ctypes.c_void_p)
    if "*" in decl:
        base = decl.replace("const", "").split("*")[0].strip()
        if base == "int":
            return ctypes.POINTER(ctypes.c_int)
        if base == "char":
            return ctypes.c_char_p
        return ctypes.c_void_p
    base = decl.replace("const", "").split()[0]
    return _CTYPE[base]


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """name -> (restype, argtypes) for every `pulpo_*` prototype declared in the header"""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)          # preprocessor lines
    text = text.replace('extern "C" {', " ")
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(pulpo_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).split(";")[-1].split("}")[-1].strip(), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else _CTYPE[ret.replace("const", "").strip()]
        argtypes = [] if args in ("", "void") else [_arg_type(a) for a in args.split(",")]
        protos[name] = (restype, argtypes)
    return protos


def header_abi_version(path: str = HEADER) -> int:
    """PULPO_ABI_VERSION of include/pulpo_hip.h (the version this binding was written against)"""
    m = re.search(r"^\s*#\s*define\s+PULPO_ABI_VERSION\s+(\d+)", open(path).read(), flags=re.M)
    if m is None:
        raise RuntimeError(f"{path} does not define PULPO_ABI_VERSION")
    return int(m.group(1))


class PulpoHipError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        self._dll = None
        self._fn = {}

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise PulpoHipError(
                f"{LIB_PATH} is missing: build it with `python -m pulpo_amd.build` (hipcc, gfx950). "
                "pulpo_amd has no CPU fallback.")
        dll = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            try:
                f = getattr(dll, name)
            except AttributeError as e:
                raise PulpoHipError(f"{LIB_PATH} does not export {name} (declared in include/pulpo_hip.h)") from e
            f.restype = restype
            f.argtypes = argtypes
            self._fn[name] = f
        self._dll = dll
        built, want = self._fn["pulpo_abi_version"](), header_abi_version()
        if built != want:
            self._dll, self._fn = None, {}
            raise PulpoHipError(f"ABI version mismatch: {LIB_PATH} was built as version {built}, include/pulpo_hip.h is version {want} "
                                "(rebuild with `python -m pulpo_amd.build --force`)")
        return dll

    def raw(self, name: str):
        self.load()
        return self._fn[name]

    def call(self, name: str, *args):
        """call an int-returning entry point and raise on a non-zero code"""
        rc = self.raw(name)(*args)
        if rc != 0:
            msg = self._fn["pulpo_last_error"]()
            raise PulpoHipError(f"{name} failed with code {rc}: {msg.decode() if msg else ''}")

    def query(self, name: str, *args):
        """call a size/count query (returns its value)"""
        return self.raw(name)(*args)


lib = _Lib()


def available() -> bool:
    return os.path.exists(LIB_PATH)
