"""Loader for libmifft.so (the hand-written HIP library behind the C ABI).

There is no fallback: if the shared library is missing or does not export every
symbol of include/mifft.h, importing the product fails loudly.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIFFT_LIBRARY points the loader at another build of the same library (A/B runs of kernel variants); it is still a
# libmifft.so and still has to export every entry point.
LIB_PATH = os.environ.get("MIFFT_LIBRARY") or os.path.join(_HERE, "csrc", "libmifft.so")

# every entry point include/mifft.h declares
EXPORTS = (
    "mifft_plan_create", "mifft_exec", "mifft_exec_batch", "mifft_plan_destroy", "mifft_plan_stages",
    "mifft_plan_kernel_name", "mifft_plan_num_launches", "mifft_plan_in_bytes", "mifft_plan_out_bytes",
    "mifft_ordered_bases", "mifft_estimate_bases", "mifft_last_error", "mifft_status_string",
    "mifft_version", "mifft_device_count", "mifft_time_exec", "mifft_jit_precompile", "mifft_plan_scratch_bytes",
    "mifft_plan_device_status", "mifft_plan_create_slab",
)


class MifftError(RuntimeError):
    """A libmifft call failed; .status is the negative mifft_status code."""

    def __init__(self, status: int, message: str):
        super().__init__(f"mifft error {status}: {message}")
        self.status = status
        self.message = message


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C hackathon_fft_amd/csrc`. There is no CPU/PyTorch fallback.")
    L = ctypes.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export {missing}")
    c = ctypes
    vp, i64, u32p, i32p = c.c_void_p, c.c_int64, c.POINTER(c.c_uint32), c.POINTER(c.c_int32)
    L.mifft_plan_create.argtypes = [c.POINTER(vp), c.c_int, c.c_int, c.c_int, c.c_int, c.POINTER(i64), i64,
                                    c.c_int, c.c_int, u32p, i32p, c.c_uint32]
    L.mifft_plan_create.restype = c.c_int
    L.mifft_plan_create_slab.argtypes = [c.POINTER(vp), c.c_int, c.c_int, c.c_int, c.c_int, c.POINTER(i64), i64,
                                         c.c_int, c.c_int, u32p, i32p, c.c_uint32, i64]
    L.mifft_plan_create_slab.restype = c.c_int
    L.mifft_exec.argtypes = [vp, vp, vp, vp]
    L.mifft_exec_batch.argtypes = [vp, vp, vp, i64, i64, vp]
    L.mifft_plan_destroy.argtypes = [vp]
    L.mifft_plan_destroy.restype = None
    L.mifft_plan_stages.argtypes = [vp, c.c_int, u32p, c.c_int]
    L.mifft_plan_kernel_name.argtypes = [vp, c.c_int]
    L.mifft_plan_kernel_name.restype = c.c_char_p
    L.mifft_plan_num_launches.argtypes = [vp]
    L.mifft_plan_in_bytes.argtypes = [vp]
    L.mifft_plan_in_bytes.restype = c.c_size_t
    L.mifft_plan_out_bytes.argtypes = [vp]
    L.mifft_plan_out_bytes.restype = c.c_size_t
    L.mifft_plan_scratch_bytes.argtypes = [vp]
    L.mifft_plan_scratch_bytes.restype = c.c_size_t
    L.mifft_plan_device_status.argtypes = [vp, vp, c.POINTER(c.c_uint32)]
    L.mifft_ordered_bases.argtypes = [c.c_uint32, u32p, c.c_int, u32p, c.c_int]
    L.mifft_estimate_bases.argtypes = [c.c_uint32, c.c_int, u32p, c.c_int]
    L.mifft_last_error.restype = c.c_char_p
    L.mifft_status_string.argtypes = [c.c_int]
    L.mifft_status_string.restype = c.c_char_p
    L.mifft_time_exec.argtypes = [vp, vp, vp, vp, c.c_int, c.POINTER(c.c_float)]
    L.mifft_jit_precompile.argtypes = [c.c_int, c.c_int, i64, c.c_int, c.c_int, c.POINTER(c.c_size_t)]
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise MifftError(rc, lib().mifft_last_error().decode(errors="replace"))
    return rc
