"""ctypes binding of libgpde.so (the C ABI declared in include/gpde.h).

The library is loaded lazily and never stored on a module/`nn.Module` instance, so models stay
picklable (`torch.save(model)`, /root/reference/graph-neural-operator/UAI1_full_resolution.py:317).
There is NO fallback: if the shared object is missing or a symbol is absent this raises.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPDE_LIB", os.path.join(_PKG, "libgpde.so"))   # GPDE_LIB: experiments

GPDE_OK = 0
GPDE_VERSION, GPDE_VERSION_ABLATION, GPDE_VERSION_INSTRUMENTED = 101, 0x10000, 0x20000
GPDE_ACC_EDGE_WEIGHTS, GPDE_ACC_ROOT, GPDE_ACC_BIAS = 1, 2, 4      # include/gpde.h (gpde_nnconv_bwd_edgeweights_acc)
GPDE_AGGR_ADD, GPDE_AGGR_MEAN, GPDE_AGGR_MAX = 0, 1, 2
GPDE_WECONV_MAX_GROUP = 16
GPDE_FWD_DEFAULT, GPDE_FWD_F16SPLIT = 0, 1
# the other forward flags of include/gpde.h (A/B switches; tests/test_abi.py checks these values against the header)
GPDE_FWD_F16SPLIT_8WAVE, GPDE_FWD_STATIC_RANGES, GPDE_FWD_AGG_F16, GPDE_FWD_AGG_F32, GPDE_FWD_NO_EDGE_PATH = 2, 4, 16, 32, 64
GPDE_WIDTH = 64
GPDE_BWD_ACCUMULATE_GRAD_HIDDEN = 1        # gpde_nnconv_bwd `flags` (tests/test_abi.py checks the value against the header)

class GpdeWeConvDesc(ctypes.Structure):
    """include/gpde.h `GpdeWeConvDesc`: one NNConv call given its per-edge weights (tests/test_abi.py checks the layout)."""
    _fields_ = [("x", ctypes.c_void_p), ("edge_weights", ctypes.c_void_p), ("rowptr", ctypes.c_void_p),
                ("src", ctypes.c_void_p), ("root", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("residual", ctypes.c_void_p), ("out", ctypes.c_void_p), ("n_nodes", ctypes.c_int32),
                ("aggr", ctypes.c_int32), ("relu", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class GpdeNodeAttr(ctypes.Structure):
    """include/gpde.h `GpdeNodeAttr`: edge attributes described by a node table (tests/test_abi.py checks the layout)."""
    _fields_ = [("table", ctypes.c_void_p), ("stride", ctypes.c_int32), ("n_slots", ctypes.c_int32), ("sel", ctypes.c_int32 * 8)]


HEADER_PATH = os.environ.get("GPDE_HEADER", os.path.join(os.path.dirname(_PKG), "include", "gpde.h"))
_SCALARS = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint32_t": ctypes.c_uint32,
            "size_t": ctypes.c_size_t, "double": ctypes.c_double}


def header_prototypes(path: str = None) -> dict:
    """{name: (return C type, [argument C types])} of every `GPDE_API` prototype of include/gpde.h (comments stripped,
    parameter names dropped, `T *` written `T*`)."""
    src = open(path or HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"GPDE_API\s+([\w \*]+?)\s*\b(gpde_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        types = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                t = re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") else a       # drop the parameter name
                types.append(t.replace(" *", "*"))
        out[name] = (ret.replace(" *", "*"), types)
    return out


def _ctype(t: str, is_return: bool = False):
    """The ctypes type of one C type of the header: scalars exactly; `const char*` results as bytes; every other pointer -
    device arrays handed over as `tensor.data_ptr()` integers, host arrays, descriptors by reference, NULL - as `void*`."""
    if t.endswith("*"):
        return ctypes.c_char_p if is_return and t == "const char*" else ctypes.c_void_p
    base = t.replace("const ", "").strip()
    if base not in _SCALARS:
        raise GpdeError(f"{HEADER_PATH}: C type {t!r} has no binding rule (graph-pde_amd/_lib.py:_ctype)")
    return _SCALARS[base]


def _signatures_from_header() -> dict:
    """name -> (restype, argtypes), GENERATED from include/gpde.h: the header is the one statement of the ABI (round 4 kept a
    hand-written mirror here, which is how 58 entry points became hard to follow; tests/test_abi.py checks the generated table
    against `nm -D` of the library and the scalar mapping against the header text)."""
    if not os.path.exists(HEADER_PATH):
        raise GpdeError(f"{HEADER_PATH} not found: the binding is generated from the header (GPDE_HEADER overrides the path)")
    return {name: (_ctype(ret, True), [_ctype(a) for a in args]) for name, (ret, args) in header_prototypes().items()}


PROF_KINDS = ("fused", "gemm3", "epilogue", "prep", "other")        # GPDE_PROF_* of include/gpde.h


def profile_begin():
    lib().gpde_profile_begin()


def profile_end():
    """{kind: (ms, launches)} of the kernels gpde_nnconv_fwd launched on this thread since profile_begin()."""
    ms = (ctypes.c_double * len(PROF_KINDS))()
    n = (ctypes.c_int32 * len(PROF_KINDS))()
    check(lib().gpde_profile_end_kinds(ms, n), "gpde_profile_end_kinds")
    return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(PROF_KINDS)}

_lib = None
_lock = threading.Lock()
n_native_calls = 0          # incremented by every gpde_nnconv_fwd call (tests assert it moves)


class GpdeError(RuntimeError):
    pass


SIGNATURES = _signatures_from_header()


def lib() -> ctypes.CDLL:
    """Load libgpde.so (once). Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise GpdeError(
                        f"{LIB_PATH} not found - build it with `python graph-pde_amd/build.py` "
                        "(or __graft_entry__.build()); the NNConv hot path has no non-HIP fallback")
                l = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(l, name)          # AttributeError if a declared symbol is missing
                    fn.restype, fn.argtypes = res, args
                ver = int(l.gpde_version())
                # the binding was generated from HEADER_PATH: a library built from another header generation has the same symbol
                # names and other argument lists - every pointer is a void* to ctypes, nothing else would notice
                m = re.search(r"#define\s+GPDE_VERSION\s+(\d+)", open(HEADER_PATH).read())
                if m is None or int(m.group(1)) != (ver & 0xffff):
                    raise GpdeError(f"{LIB_PATH} reports ABI version {ver & 0xffff}, {HEADER_PATH} declares "
                                    f"{m.group(1) if m else '?'}: header and library do not belong together (GPDE_LIB / GPDE_HEADER)")
                if ver & GPDE_VERSION_ABLATION and os.environ.get("GPDE_ALLOW_ABLATION") != "1":
                    raise GpdeError(
                        f"{LIB_PATH} is an ABLATION build (gpde_version() = {ver:#x}: arithmetic compiled out, results are "
                        "wrong; timing experiments only) - refused.  GPDE_ALLOW_ABLATION=1 loads it for such an experiment")
                _lib = l
    return _lib


def reload_switches() -> None:
    """Re-read the library's GPDE_* developer switches from the environment (they are read once per process otherwise)."""
    if _lib is not None:
        _lib.gpde_reload_switches()


def check(rc: int, what: str) -> None:
    if rc != GPDE_OK:
        msg = lib().gpde_last_error().decode("utf-8", "replace")
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        raise GpdeError(f"{what} failed (code {rc}): {msg}")


def dims_array(dims):
    return (ctypes.c_int32 * len(dims))(*[int(d) for d in dims])
