"""ctypes binding of libmssvt_hip.so (include/mssvt_hip.h).

The product path has NO CPU fallback: if the HIP library is missing this module
raises at first use.  Arguments are raw device pointers (``tensor.data_ptr()``)
plus the current torch HIP stream -- torch is only the allocator / stream owner.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSSVT_LIB: another build of the same library (the schedule variants of mssvt_amd/build.py, for the hazard test)
LIB_PATH = os.environ.get("MSSVT_LIB") or os.path.join(_HERE, "lib", "libmssvt_hip.so")

_lib = None


class MssvtHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MssvtHipError(
                "libmssvt_hip.so not built (%s). Run `python -m mssvt_amd.build`; there is no "
                "CPU fallback for the MsSVT hot path." % LIB_PATH)
        lib_ = ctypes.CDLL(LIB_PATH)
        lib_.mssvt_hip_status_string.restype = ctypes.c_char_p
        for name in ("mssvt_hash_workspace_ints", "mssvt_nms_workspace_bytes", "mssvt_linear_wgrad_workspace_floats",
                     "mssvt_csr_transpose_workspace_bytes", "mssvt_ffn_packed_bytes", "mssvt_level_sorted_scratch_ints",
                     "mssvt_attn_packed_bytes", "mssvt_frame_workspace_bytes", "mssvt_compress_ws_packed_bytes",
                     "mssvt_train_tok_slab_floats"):
            getattr(lib_, name).restype = ctypes.c_longlong
        global TYPED
        TYPED = _declare(lib_)
        _lib = lib_  # only a fully declared library is cached
    return _lib


_NULL = ctypes.c_void_p(0)
TYPED = False  # argtypes of every entry point set from include/mssvt_hip.h: callers may pass plain ints / floats / addresses


def _declare(lib_):
    """argtypes / restype of every `mssvt_*` function from the declarations of include/mssvt_hip.h (int, float,
    long long, pointers).  With them ctypes converts plain Python ints and floats in C: the fused forward passes ~600
    scalar / pointer arguments per frame, and a ctypes.c_int / c_void_p object per argument cost ~0.1 ms of host time per
    frame.  Without the header (a stripped install) nothing is declared and the callers keep wrapping."""
    import re
    path = os.path.join(os.path.dirname(_HERE), "include", "mssvt_hip.h")
    if not os.path.exists(path):
        return False
    with open(path) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    kinds = {"int": ctypes.c_int, "float": ctypes.c_float, "long long": ctypes.c_longlong}
    rets = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "const char *": ctypes.c_char_p}
    for ret, name, params in re.findall(r"\b(int|long long|const char \*)\s*(mssvt_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        fn = getattr(lib_, name, None)
        if fn is None:
            continue
        params = params.replace("\n", " ").strip()
        args = []
        for q in ([] if params in ("", "void") else params.split(",")):
            q = q.strip()
            args.append(ctypes.c_void_p if "*" in q else kinds.get(" ".join(q.split()[:-1])))
        if None in args:
            continue  # a parameter type this parser does not know: leave the function undeclared (callers wrap)
        fn.argtypes = args
        fn.restype = rets[ret]
    return True


def check(status, what):
    if status != 0:
        raise MssvtHipError("%s failed: %s (status %d)" % (
            what, lib().mssvt_hip_status_string(int(status)).decode(), status))


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return _NULL
    if not (t.is_cuda and t.is_contiguous()):
        raise MssvtHipError("mssvt_amd ops need contiguous tensors on the GPU (no CPU path)")
    return ctypes.c_void_p(t.data_ptr())


def ptr_fast(t):
    """`ptr` without the device / contiguity checks: for buffers the fused path allocated itself and for module
    parameters (the frame's front is launch bound: ~200 pointer conversions per frame)."""
    return _NULL if t is None else ctypes.c_void_p(t.data_ptr())


def ptr_raw(t):
    """Address of a device buffer as a plain int (None -> NULL): for entry points with declared argtypes."""
    return None if t is None else t.data_ptr()


def stream():
    """Raw handle of torch's current HIP stream (the fast C accessor: torch.cuda.current_stream()
    builds a Python Stream object, ~8 us per call, 20+ calls per forward)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


_fns = {}


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(lib(), name)
    status = fn(*args)
    if status != 0:
        check(status, name)
