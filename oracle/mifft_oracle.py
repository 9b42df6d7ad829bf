"""ctypes wrapper over the CPU oracle (oracle/libmifft_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package.

Mirrors the reference's CPU call surface: ``plan_fft(...) -> plan`` and
``fft(output, x, plan=plan, cpu_workers=None)`` (fft/fft/fft.mojo:123-157,
213-259) on numpy arrays laid out ``(batch, d0[, d1[, d2]], C)``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmifft_oracle.so")

_DTYPES = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.uint8): 2, np.dtype(np.int32): 3,
           np.dtype(np.int8): 4, np.dtype(np.int16): 5, np.dtype(np.uint16): 6, np.dtype(np.float16): 7}
BF16 = "bfloat16"  # pass in_dtype=BF16 with a uint16 array of bfloat16 bit patterns (numpy has no bfloat16)


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (g++)."""
    src = os.path.join(_HERE, "mifft_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.mifft_oracle_last_error.restype = ctypes.c_char_p
        L.mifft_oracle_plan_create.argtypes = [
            ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, ctypes.c_int, ctypes.c_int,
            ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
        L.mifft_oracle_exec.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.mifft_oracle_exec_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        L.mifft_oracle_plan_destroy.argtypes = [ctypes.c_void_p]
        L.mifft_oracle_plan_stages.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
        L.mifft_oracle_ordered_bases.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
        L.mifft_oracle_estimate_bases.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_int]
        _lib = L
    return _lib


class OracleError(ValueError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"[{status}] {msg}")
        self.status = status


def _check(rc: int) -> None:
    if rc < 0:
        raise OracleError(rc, lib().mifft_oracle_last_error().decode())


def ordered_bases(length: int, bases: Sequence[int]) -> list:
    """_get_ordered_bases_processed_list()[0], fft/fft/_utils.mojo:186-221."""
    arr = (ctypes.c_uint32 * max(len(bases), 1))(*bases)
    out = (ctypes.c_uint32 * 64)()
    n = lib().mifft_oracle_ordered_bases(length, arr, len(bases), out, 64)
    _check(n)
    return list(out[:n])


def estimate_bases(length: int, target: str = "cpu") -> list:
    """_estimate_best_bases, fft/fft/fft.mojo:49-104."""
    out = (ctypes.c_uint32 * 64)()
    n = lib().mifft_oracle_estimate_bases(length, 1 if target == "gpu" else 0, out, 64)
    _check(n)
    return list(out[:n])


class OraclePlan:
    """_CPUPlan (fft/fft/_ndim_fft_cpu.mojo:28-60)."""

    def __init__(self, in_dtype, out_dtype, in_shape: Sequence[int], out_shape: Sequence[int], *,
                 inverse: bool = False, bases: Optional[Sequence[Sequence[int]]] = None,
                 default_target: str = "cpu"):
        in_shape, out_shape = tuple(in_shape), tuple(out_shape)
        # _check_layout_conditions_nd, fft/fft/fft.mojo:20-46
        if len(out_shape) <= 2:
            raise OracleError(-1, "The rank should be bigger than 2.")
        if len(in_shape) != len(out_shape):
            raise OracleError(-1, "in_layout and out_layout must have equal rank")
        if out_shape[-1] != 2:
            raise OracleError(-3, "out_layout must have the last dimension equal to 2")
        if in_shape[:-1] != out_shape[:-1]:
            raise OracleError(-2, "out_layout and in_layout should have the same shape before the last dimension")
        self.bf16 = isinstance(in_dtype, str) and in_dtype == BF16
        self.in_dtype, self.out_dtype = np.dtype(np.uint16 if self.bf16 else in_dtype), np.dtype(out_dtype)
        self.in_shape, self.out_shape = in_shape, out_shape
        self.inverse = bool(inverse)
        dims = out_shape[1:-1]
        if bases is not None and len(bases) != len(dims):
            raise OracleError(-7, "The bases list should have the same outer size as the amount of internal dimensions.")
        c_dims = (ctypes.c_int64 * len(dims))(*dims)
        if bases is not None:
            flat = [int(b) for bs in bases for b in bs]
            c_flat = (ctypes.c_uint32 * max(len(flat), 1))(*flat)
            c_len = (ctypes.c_int32 * len(dims))(*[len(bs) for bs in bases])
        else:
            c_flat, c_len = None, None
        h = ctypes.c_void_p()
        if self.in_dtype not in _DTYPES or self.out_dtype not in _DTYPES:
            raise OracleError(-4, "unsupported dtype")
        rc = lib().mifft_oracle_plan_create(ctypes.byref(h), 8 if self.bf16 else _DTYPES[self.in_dtype], _DTYPES[self.out_dtype],
                                            len(dims), c_dims, out_shape[0], in_shape[-1], int(self.inverse),
                                            c_flat, c_len, 1 if default_target == "gpu" else 0)
        _check(rc)
        self._h = h

    def stages(self, dim: int) -> list:
        out = (ctypes.c_uint32 * 64)()
        n = lib().mifft_oracle_plan_stages(self._h, dim, out, 64)
        _check(n)
        return list(out[:n])

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib().mifft_oracle_plan_destroy(h)


def plan_fft(in_dtype, out_dtype, in_shape, out_shape, *, inverse=False, bases=None,
             cpu_workers=None, default_target="cpu") -> OraclePlan:
    return OraclePlan(in_dtype, out_dtype, in_shape, out_shape, inverse=inverse, bases=bases,
                      default_target=default_target)


def fft(output: np.ndarray, x: np.ndarray, *, plan: OraclePlan, cpu_workers: Optional[int] = None,
        first: int = 0, count: Optional[int] = None) -> None:
    assert output.flags.c_contiguous and x.flags.c_contiguous
    assert output.dtype == plan.out_dtype and x.dtype == plan.in_dtype
    assert tuple(output.shape) == plan.out_shape and tuple(x.shape) == plan.in_shape
    if count is None:
        count = plan.out_shape[0] - first
    rc = lib().mifft_oracle_exec_batch(plan._h, x.ctypes.data, output.ctypes.data, first, count,
                                       int(cpu_workers or 0))
    _check(rc)


def fftn(x: np.ndarray, *, inverse=False, bases=None, out_dtype=None, cpu_workers=None, in_dtype=None) -> np.ndarray:
    """Convenience: x is (batch, d0.., C) real-typed, or complex (batch, d0..).  ``in_dtype=BF16``: x holds bfloat16 bit
    patterns as uint16."""
    if np.iscomplexobj(x):
        x = np.ascontiguousarray(x)
        x = x.view(x.real.dtype).reshape(x.shape + (2,))
    x = np.ascontiguousarray(x)
    if out_dtype is None:
        out_dtype = x.dtype if x.dtype in (np.float32, np.float64) else np.float64
    out_shape = x.shape[:-1] + (2,)
    plan = plan_fft(in_dtype if in_dtype is not None else x.dtype, out_dtype, x.shape, out_shape, inverse=inverse, bases=bases)
    out = np.full(out_shape, np.nan, dtype=out_dtype)
    fft(out, x, plan=plan, cpu_workers=cpu_workers)
    return out


def num_procs() -> int:
    return int(lib().mifft_oracle_num_procs())
