"""ctypes loader of the C-ABI kernel library `liblvc_amd.so` (declared in include/lvc_amd.h).

The product path has NO CPU fallback: if the library is missing or a call fails, this raises.
Errors follow the reference's behaviour for its native ops (C++ exception -> Python RuntimeError,
reference detectron2/layers/csrc/ROIAlign/ROIAlign_cuda.cu:318-324): every C function returns an
int status, non-zero is re-raised here as RuntimeError carrying `lvc_last_error()`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LVC_AMD_LIB: another build of the same library (A/B measurements of one kernel against its predecessor in alternating processes)
_SO = os.environ.get("LVC_AMD_LIB") or os.path.join(_HERE, "liblvc_amd.so")
_lib = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_double = ctypes.c_double
c_longlong = ctypes.c_longlong


class LvcNativeError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError(
                "lvc_amd: native library {} not found; build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C lvc_amd/csrc` "
                "(there is no CPU fallback)".format(_SO)
            )
        L = ctypes.CDLL(_SO)
        L.lvc_last_error.restype = ctypes.c_char_p
        L.lvc_batched_nms_workspace_bytes.restype = c_longlong
        L.lvc_abi_version.restype = c_int
        L.lvc_set_wino_streamk.restype = None
        if os.environ.get("LVC_WINO_STREAMK", "0") != "0":
            L.lvc_set_wino_streamk(c_int(int(os.environ["LVC_WINO_STREAMK"])))
        L.lvc_set_nms_reduce_global.restype = None
        L.lvc_set_halo_test_hooks.restype = None
        if os.environ.get("LVC_NMS_REDUCE_GLOBAL", "0") != "0":      # read once, here: the library itself never looks at the environment
            L.lvc_set_nms_reduce_global(c_int(1))
        _lib = L
    return _lib


def check(status, what=""):
    if status != 0:
        msg = lib().lvc_last_error().decode("utf-8", "replace")
        # a failed launch must not leave the thread's range slot on the failing layer (it is reset on the success path only):
        # later un-slotted launches would raise that layer's word instead of the shared one
        lib().lvc_set_range_slot(0)
        raise LvcNativeError("{} failed (status {}): {}".format(what or "lvc_amd native call", status, msg))


def ptr(t):
    """Device (or host) pointer of a tensor, or NULL for None."""
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch

    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
