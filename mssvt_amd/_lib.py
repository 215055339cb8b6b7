"""ctypes binding of libmssvt_hip.so (include/mssvt_hip.h).

The product path has NO CPU fallback: if the HIP library is missing this module
raises at first use.  Arguments are raw device pointers (``tensor.data_ptr()``)
plus the current torch HIP stream -- torch is only the allocator / stream owner.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmssvt_hip.so")

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
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.mssvt_hip_status_string.restype = ctypes.c_char_p
        _lib.mssvt_hash_workspace_ints.restype = ctypes.c_longlong
    return _lib


def check(status, what):
    if status != 0:
        raise MssvtHipError("%s failed: %s (status %d)" % (
            what, lib().mssvt_hip_status_string(int(status)).decode(), status))


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_cuda, "mssvt_amd ops need tensors on the GPU (no CPU path)"
    assert t.is_contiguous(), "mssvt_amd ops need contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    fn = getattr(lib(), name)
    check(fn(*args), name)
