"""ctypes binding of libagent0_hip.so (the C-ABI declared in include/agent0_hip.h).

The prototypes are parsed from the header itself, so the Python side can never
drift from the ABI; a missing library or a missing symbol is a hard error —
there is no CPU fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, List, Tuple

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(ROOT, "include", "agent0_hip.h")
LIB_PATH = os.path.join(_PKG, "lib", "libagent0_hip.so")

_SCALARS = {
    "int": C.c_int,
    "unsigned int": C.c_uint,
    "long long": C.c_longlong,
    "unsigned long long": C.c_ulonglong,
    "float": C.c_float,
    "double": C.c_double,
}


class A0Error(RuntimeError):
    pass


class NetDesc(C.Structure):
    _fields_ = [("C", C.c_int), ("H", C.c_int), ("W", C.c_int)]


class FramesArg(C.Structure):
    _fields_ = [("frames", C.c_void_p), ("slot", C.c_void_p), ("sample_stride", C.c_longlong), ("chan_off", C.c_int)]


class EncoderWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w1", "b1", "w2", "b2", "w3", "b3")]


class LearnerDesc(C.Structure):
    _fields_ = [("A", C.c_int), ("dueling", C.c_int), ("double_q", C.c_int), ("B", C.c_int), ("n_step", C.c_int), ("discount", C.c_double), ("lr", C.c_double),
                ("adam_eps", C.c_double), ("target_update_freq", C.c_int), ("algo", C.c_int), ("num_atoms", C.c_int), ("vmin", C.c_double), ("vmax", C.c_double),
                ("noisy", C.c_int), ("seed", C.c_ulonglong), ("iqn_K", C.c_int), ("iqn_N", C.c_int), ("iqn_N_dash", C.c_int), ("fqf_F", C.c_int), ("mdqn_tau", C.c_double), ("mdqn_lo", C.c_double),
                ("max_grad_norm", C.c_double)]


class ReduceSeg(C.Structure):
    _fields_ = [("slabs", C.c_void_p), ("slab_stride", C.c_longlong), ("nslab", C.c_int), ("out", C.c_void_p), ("count", C.c_longlong)]


class PendingReduce(C.Structure):
    _fields_ = [("seg", ReduceSeg * 4), ("n", C.c_int)]


class EncoderPass(C.Structure):
    _fields_ = [("wt", C.c_void_p), ("w", C.c_void_p), ("f", C.c_void_p), ("B", C.c_int), ("act1", C.c_void_p), ("act2", C.c_void_p), ("act3", C.c_void_p)]


def parse_header(path: str = HEADER) -> List[Tuple[str, str, List[str]]]:
    """-> [(return type, name, [arg types])] for every function prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"typedef struct \w+ \{.*?\} \w+;", " ", text, flags=re.S)
    protos = []
    for m in re.finditer(r"(const char\*|long long|int)\s+(a0_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        types: List[str] = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append("ptr")
                else:
                    t = re.sub(r"\b\w+$", "", a).strip()  # drop the parameter name
                    t = t.replace("const ", "").strip()
                    if t not in _SCALARS:
                        raise ValueError(f"unhandled C type {t!r} in {name}")
                    types.append(t)
        protos.append((ret, name, types))
    return protos


_lib = None
_protos: Dict[str, Tuple[str, List[str]]] = {}


def load(path: str = LIB_PATH, allow_variant: bool = False):
    """Load the shared library and attach argtypes/restype.  Raises if anything is missing.

    The product loads exactly one file, ``agent0_amd/lib/libagent0_hip.so``, and only a build that reports itself as "default"
    (``a0_build_info``).  Tuning builds (tools/build_variant.sh: extra -D flags, possibly timing-only code) are loaded by the
    diagnostics under ``tools/`` alone, through ``tools/with_lib.py``, which passes ``allow_variant=True`` before anything else loads."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise A0Error(
            f"{path} not found: agent0_amd needs its HIP library (build it with agent0_amd/csrc/build.sh or "
            f"python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback."
        )
    lib = C.CDLL(path)
    for ret, name, types in parse_header():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise A0Error(f"libagent0_hip.so does not export {name} (declared in include/agent0_hip.h)") from e
        fn.restype = {"int": C.c_int, "long long": C.c_longlong, "const char*": C.c_char_p}[ret]
        fn.argtypes = [C.c_void_p if t == "ptr" else _SCALARS[t] for t in types]
        _protos[name] = (ret, types)
    if lib.a0_abi_version() != 1:
        raise A0Error("libagent0_hip.so ABI version mismatch")
    info = (lib.a0_build_info() or b"").decode()
    if info != "default" and not allow_variant:
        raise A0Error(f"{path} is a tuning build ({info}); the product only loads the default build of agent0_amd/csrc/build.sh")
    _lib = lib
    return lib


def last_error() -> str:
    return (load().a0_last_error() or b"").decode()


def check(status: int, what: str = ""):
    if status != 0:
        raise A0Error(f"{what or 'libagent0_hip'} failed with status {status}: {last_error()}")


def ptr(t) -> int | None:
    """Device (or host) address of a torch tensor / None."""
    if t is None:
        return None
    return t.data_ptr()
