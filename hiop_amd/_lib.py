"""ctypes loader for libhiopamd.so.

The prototypes are taken from include/hiop_amd.h itself (parsed once at import), so the Python side
can never drift from the C ABI.  Loading fails loudly when the HIP library has not been built; there
is no Python/CPU fallback for any entry point.
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent
HEADER = ROOT.parent / "include" / "hiop_amd.h"
import os as _os
_VARIANT = _os.environ.get("HIOPAMD_BUILD_VARIANT", "")          # a test build of hiop_amd/build.py (e.g. "poison"); default: the shipped one
LIBPATH = ROOT / ("lib" if not _VARIANT else f"lib_{_VARIANT}") / "libhiopamd.so"

_OPAQUE = {"hiopamd_ctx", "hiopamd_sp_plan", "hiopamd_linsolver", "hiopamd_kkt_mds", "hiopamd_kkt_lowrank",
           "hiopamd_hess_lowrank"}


class MdsStructure(C.Structure):
    _fields_ = [
        ("nxs", C.c_int), ("nxd", C.c_int), ("neq", C.c_int), ("nineq", C.c_int),
        ("nnz_Jcs", C.c_int), ("Jcs_i", C.c_void_p), ("Jcs_j", C.c_void_p), ("Jcs_i_host", C.c_void_p),
        ("Jcs_j_host", C.c_void_p),
        ("nnz_Jds", C.c_int), ("Jds_i", C.c_void_p), ("Jds_j", C.c_void_p), ("Jds_i_host", C.c_void_p),
        ("Jds_j_host", C.c_void_p),
        ("nnz_Hss", C.c_int), ("Hss_i", C.c_void_p), ("Hss_j", C.c_void_p),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
LINOP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)   # hiopamd_linop_fn(user, x_dev, y_dev)


def _ctype(t: str):
    t = t.replace("const", "").strip()
    t = re.sub(r"\s+", " ", t)
    if t.endswith("*"):
        return C.c_void_p  # all pointers travel as raw addresses
    if t in ("int", "hiopamd_status", "hiopamd_redop"):
        return C.c_int
    if t == "int64_t":
        return C.c_int64
    if t == "uint64_t":
        return C.c_uint64
    if t == "size_t":
        return C.c_size_t
    if t == "double":
        return C.c_double
    if t == "void":
        return None
    if t == "hiopamd_allreduce_fn":
        return ALLREDUCE_FN
    if t == "hiopamd_linop_fn":
        return LINOP_FN
    raise ValueError(f"unmapped C type {t!r}")


def parse_header(path: Path = HEADER):
    """Return {name: (restype, [argtypes])} for every function the header declares."""
    txt = path.read_text()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    txt = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", txt, flags=re.S)
    txt = re.sub(r"typedef\s+enum\s*\{.*?\}\s*\w+\s*;", " ", txt, flags=re.S)
    txt = re.sub(r"typedef[^;]*;", " ", txt)
    txt = re.sub(r"^\s*#.*$", " ", txt, flags=re.M)          # preprocessor lines
    txt = re.sub(r'extern\s+"C"\s*\{', " ", txt)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(hiopamd_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name (if any)
                mm = re.match(r"^(.*?[\*\s])(\w+)$", a)
                if mm and mm.group(2) not in ("int", "double", "int64_t", "size_t", "void") and \
                        not mm.group(2).startswith("hiopamd_"):
                    a = mm.group(1)
                argtypes.append(_ctype(a.strip()))
        protos[name] = (_ctype(ret), argtypes)
    return protos


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIBPATH.exists():
            raise RuntimeError(
                f"{LIBPATH} is missing: build it with `python -m hiop_amd.build` (hipcc, gfx950). "
                "hiop_amd has no CPU fallback.")
        # PyTorch ships its own libamdhip64.  If libhiopamd.so were loaded first it would pull in /opt/rocm's copy and a
        # later `import torch` would bring a second HIP runtime into the process (double free at exit): torch goes first,
        # libhiopamd.so then binds to the runtime that is already there and shares torch's device memory and streams.
        import torch  # noqa: F401
        L = C.CDLL(str(LIBPATH), mode=C.RTLD_GLOBAL)
        for name, (ret, args) in parse_header().items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch: fail loudly
            fn.restype = ret
            fn.argtypes = args
        _lib = L
    return _lib


class HiopAmdError(RuntimeError):
    pass


_ERR = {-1: "HIP runtime error", -2: "invalid argument", -3: "no gfx950 device", -4: "singular", -5: "bad call sequence"}   # (-6 time-out, -7 solve failed: shown as numbers)


def check(rc: int, what: str = ""):
    if rc != 0:
        raise HiopAmdError(f"{what} failed: {_ERR.get(rc, rc)}")
