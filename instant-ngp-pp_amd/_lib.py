"""ctypes binding of libngp_hip.so (the C ABI in include/ngp_hip.h).

The prototypes are parsed from the header at import time, so the Python side cannot drift
from the declared ABI.  There is NO CPU fallback: if the library is missing and cannot be
built, or a call returns an error code, this module raises.
"""
import ctypes as C
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "ngp_hip.h")
LIB_PATH = os.path.join(HERE, "libngp_hip.so")

_CTYPES = {
    "int": C.c_int, "int64_t": C.c_int64, "float": C.c_float, "double": C.c_double,
    "uint32_t": C.c_uint32,
}


class GridDesc(C.Structure):
    """struct ngp_grid_desc"""
    _fields_ = [("n_levels", C.c_uint32), ("n_features", C.c_uint32),
                ("offsets", C.c_uint32 * 33), ("resolution", C.c_uint32 * 32), ("scale", C.c_float * 32)]


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every `ngp_*` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(ngp_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = {"int": C.c_int, "int64_t": C.c_int64, "const char*": C.c_char_p}[ret]
        argl = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                an = re.search(r"(\w+)$", a).group(1)
                ty = a[: a.rfind(an)].strip()
                if "*" in ty:
                    argl.append((C.c_void_p, an))
                else:
                    argl.append((_CTYPES[ty.replace("const ", "").strip()], an))
        protos[name] = (restype, argl)
    return protos


PROTOS = parse_header()
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # The prototypes above come from the header as it is NOW; a library built from other sources would be
    # called with the wrong argument lists.  build.py stamps the library with the digest of its sources:
    # a missing or stale library is rebuilt when hipcc is present, and refused otherwise.
    from . import build as _build
    override = os.environ.get("NGP_LIB_OVERRIDE") if os.environ.get("NGP_AB_VARIANTS") else None
    if override:
        # A/B experiments only (honoured with NGP_AB_VARIANTS=1): a library built from ANOTHER revision with the same
        # header, for same-box comparisons of two kernel versions; the build-id check is skipped, the symbol check is not
        import sys
        print(f"[ngp_amd] NGP_LIB_OVERRIDE: loading {override} without the build-id check", file=sys.stderr)
        lib = C.CDLL(override)
        for name, (restype, argl) in PROTOS.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = [t for t, _ in argl]
        _lib = lib
        return lib
    want = _build.source_id()
    if not os.path.exists(LIB_PATH) or _build.built_id() != want:
        if not _build.have_hipcc():
            raise RuntimeError(f"{LIB_PATH} is missing or was built from other sources (id {_build.built_id()}, "
                               f"sources {want}) and hipcc is not available to rebuild it")
        _build.build()
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argl) in PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = [t for t, _ in argl]
    got = lib.ngp_build_id().decode()
    if got != want:
        raise RuntimeError(f"{LIB_PATH} reports build id {got}, its sources have {want}: rebuild it "
                           "(python -m instant-ngp-pp_amd.build)")
    _lib = lib
    return lib


def check_input(t, name="x"):
    """CHECK_INPUT of the reference (models/csrc/include/utils.h:4-6)."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _arg(v):
    if v is None:
        return None
    if isinstance(v, torch.Tensor):
        return v.data_ptr()
    if isinstance(v, C.Structure):
        return C.addressof(v)
    return v


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


# Optional per-entry-point device timing: {name: [(start_event, end_event, tag), ...]}.  bench.py sets
# this to bracket the kernels of the timed region with HIP events on the launch stream.
PROFILE = None
# Optional capture of the argument tuples of chosen entry points ({name: [args, ...]}): bench.py replays them alone
# after the timed region to report every kernel's solo rate beside its in-step rate.
CAPTURE = None


def call(name, *args):
    """Calls ngp_<name>(*args, current HIP stream).  Tensors are passed as device pointers."""
    lib = load()
    if CAPTURE is not None and name in CAPTURE:
        CAPTURE[name].append(args)
    prof = PROFILE
    if prof is not None and name in prof:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, "ngp_" + name)(*[_arg(a) for a in args], stream_ptr())
        e1.record()
        prof[name].append((e0, e1, tuple(a for a in args if isinstance(a, int))))
    else:
        rc = getattr(lib, "ngp_" + name)(*[_arg(a) for a in args], stream_ptr())
    if rc != 0:
        raise RuntimeError(f"ngp_{name} failed with code {rc}")


def call_host(name, *args):
    """Host-only entry points (no stream argument); returns the raw result."""
    return getattr(load(), "ngp_" + name)(*[_arg(a) for a in args])
