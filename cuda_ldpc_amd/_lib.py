"""Loader of libcuda_ldpc_amd.so (the C ABI declared in include/bldpc.h, include/nbldpc.h)."""
import ctypes
import os

import torch  # noqa: F401  -- first, so that ONE HIP runtime (torch's) serves the whole process

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("CUDA_LDPC_AMD_SO") or os.path.join(_HERE, "libcuda_ldpc_amd.so")  # env: experiments only


class ExtensionMissing(ImportError):
    pass


def load():
    if not os.path.exists(SO_PATH):
        raise ExtensionMissing(
            "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback." % SO_PATH)
    return ctypes.CDLL(SO_PATH)


lib = load()

c_int, c_void_p, c_char_p = ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p
lib.bldpc_last_error.restype = c_char_p
lib.bldpc_last_kernel.restype = c_char_p
lib.bldpc_last_kernel.argtypes = [c_void_p]
lib.bldpc_read_blockh.argtypes = [c_char_p, c_int, c_int, c_void_p, c_void_p, c_void_p]
lib.bldpc_transform_h.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int]
lib.bldpc_code_create_qc.argtypes = [c_int, c_int, c_int, c_void_p, ctypes.POINTER(c_void_p)]
lib.bldpc_code_create_table.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p)]
lib.bldpc_code_destroy.argtypes = [c_void_p]
lib.bldpc_code_dims.argtypes = [c_void_p, c_void_p]
lib.bldpc_decode.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                             ctypes.POINTER(c_int), c_void_p]
lib.bldpc_decode_per_frame.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
lib.bldpc_statistic.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
lib.bldpc_statistic_per_frame.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]
lib.bldpc_decode_statistic.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
lib.bldpc_set_profiling.argtypes = [c_void_p, c_int]
lib.bldpc_last_kernel_ms.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_float)]
lib.bldpc_kernel_ms_mean.argtypes = [c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(c_int)]
lib.bldpc_awgn_channel_host.argtypes = [c_void_p, ctypes.c_float, c_void_p, c_void_p, c_int, c_int]
lib.bldpc_awgn_channel_device.argtypes = [c_void_p, ctypes.c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]
lib.bldpc_sigma.restype = ctypes.c_float
lib.bldpc_sigma.argtypes = [ctypes.c_float, c_int, ctypes.c_float]


class LdpcError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise LdpcError("%s failed (%d): %s" % (what, rc, lib.bldpc_last_error().decode(errors="replace")))
