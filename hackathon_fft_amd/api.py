"""Host-side mirror of the reference's GPU call surface, over the libmifft C ABI.

Reference interface (fft/fft/fft.mojo):
    plan_fft[in_dtype, out_dtype, in_layout, out_layout, *, bases, inverse,
             runtime_twfs, max_cluster_size, _test](*, ctx) -> _GPUPlan   (:161-210)
    fft(output, x, ctx, *, plan)                                          (:262-323)
Mojo's compile-time parameters become ordinary arguments here; layouts are
shapes ``(batch, d0[, d1[, d2]], C)`` of row-major tensors.

PyTorch is used only as the owner of device memory and streams: tensors are
handed to the library as raw device pointers.  Nothing in this module computes
an FFT on the host or through torch.fft.
"""
from __future__ import annotations

import collections
import ctypes
import enum
import threading
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import MifftError, check

# mifft_dtype (include/mifft.h).  Input tensors may have any of these element types -- the reference casts in its
# first-stage load (fft/fft/_fft.mojo:243-257); the output is float32 or float64.
_DTYPE_CODE = {torch.float32: 0, torch.float64: 1, torch.uint8: 2, torch.int32: 3, torch.int8: 4, torch.int16: 5,
               torch.float16: 7, torch.bfloat16: 8}
if hasattr(torch, "uint16"):
    _DTYPE_CODE[torch.uint16] = 6
_OUT_DTYPES = (torch.float32, torch.float64)

FLAG_FAITHFUL_STAGES = 1


class GPUTest(enum.Enum):
    """The reference's code-path forcing knob (_GPUTest, fft/fft/_ndim_fft_gpu.mojo:453-459).

    CDNA4 has no thread-block clusters and one synchronisation scope that matters
    (the workgroup), so every value selects the same thing here: the literal
    stage-per-pass kernel family instead of the fused register-butterfly kernels.
    """
    BLOCK = 0
    WARP = 1
    DEVICE_WIDE = 2
    CLUSTER = 3


class DeviceContext:
    """Stand-in for Mojo's DeviceContext: a device plus the stream work is enqueued on."""

    def __init__(self, device: Optional[int] = None, stream: Optional["torch.cuda.Stream"] = None):
        if not torch.cuda.is_available():
            raise MifftError(-10, "no HIP device visible to torch; libmifft has no CPU path")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._stream = stream

    @property
    def stream(self) -> "torch.cuda.Stream":
        return self._stream if self._stream is not None else torch.cuda.current_stream(self.device)

    def synchronize(self) -> None:
        self.stream.synchronize()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def ordered_bases(length: int, bases: Sequence[int]) -> list:
    """Descending per-stage radices (_get_ordered_bases_processed_list, fft/fft/_utils.mojo:186-221)."""
    n = len(bases)
    arr = (ctypes.c_uint32 * max(n, 1))(*[int(b) for b in bases])
    out = (ctypes.c_uint32 * 64)()
    k = check(_lib.lib().mifft_ordered_bases(int(length), arr, n, out, 64))
    return list(out[:k])


def estimate_best_bases(length: int, target: str = "gpu") -> list:
    """_estimate_best_bases (fft/fft/fft.mojo:49-104)."""
    out = (ctypes.c_uint32 * 64)()
    k = check(_lib.lib().mifft_estimate_bases(int(length), 1 if target == "gpu" else 0, out, 64))
    return list(out[:k])


def estimate_best_bases_nd(in_shape: Sequence[int], out_shape: Sequence[int], target: str = "gpu") -> list:
    """_estimate_best_bases_nd (fft/fft/fft.mojo:107-119)."""
    _check_layout_conditions_nd(tuple(in_shape), tuple(out_shape))
    return [estimate_best_bases(d, target) for d in out_shape[1:-1]]


def _check_layout_conditions_nd(in_shape: tuple, out_shape: tuple) -> None:
    """_check_layout_conditions_nd (fft/fft/fft.mojo:20-46), raised at plan time instead of compile time."""
    rank = len(out_shape)
    if rank <= 2:
        raise MifftError(-1, "The rank should be bigger than 2. The first dimension represents the amount of "
                             "batches, and the last the complex dimension.")
    if len(in_shape) != rank:
        raise MifftError(-1, "in_layout and out_layout must have equal rank")
    if not 1 <= in_shape[-1] <= 2:
        raise MifftError(-3, "The last dimension of in_layout should be 1 or 2")
    if out_shape[-1] != 2:
        raise MifftError(-3, "out_layout must have the last dimension equal to 2")
    if tuple(out_shape[:-1]) != tuple(in_shape[:-1]):
        raise MifftError(-2, "out_layout and in_layout should have the same shape before the last dimension")
    for i in range(rank - 2):
        if out_shape[i + 1] == 1:
            raise MifftError(-2, "no inner dimension should be of size 1")


class Plan:
    """_GPUPlan (fft/fft/_ndim_fft_gpu.mojo:153-207): owns the device twiddle tables."""

    def __init__(self, in_dtype, out_dtype, in_shape, out_shape, *, bases=None, inverse=False,
                 device: int = 0, flags: int = 0, whole_batch: int = 0):
        in_shape, out_shape = tuple(int(v) for v in in_shape), tuple(int(v) for v in out_shape)
        _check_layout_conditions_nd(in_shape, out_shape)
        if in_dtype not in _DTYPE_CODE or out_dtype not in _OUT_DTYPES:
            raise MifftError(-4, f"unsupported dtype {in_dtype} -> {out_dtype} (out_dtype must be floating point)")
        dims = out_shape[1:-1]
        if bases is not None and len(bases) != len(dims):
            raise MifftError(-7, "The bases list should have the same outer size as the amount of internal "
                                 "dimensions. e.g. (batches, dim_0, dim_1, dim_2, 2) -> len(bases) == 3")
        self.in_dtype, self.out_dtype = in_dtype, out_dtype
        self.in_shape, self.out_shape = in_shape, out_shape
        self.inverse, self.device, self.flags = bool(inverse), int(device), int(flags)
        c_dims = (ctypes.c_int64 * len(dims))(*dims)
        if bases is not None:
            flat = [int(b) for bs in bases for b in bs]
            c_flat = (ctypes.c_uint32 * max(len(flat), 1))(*flat)
            c_len = (ctypes.c_int32 * len(dims))(*[len(bs) for bs in bases])
        else:
            c_flat, c_len = None, None
        h = ctypes.c_void_p()
        # whole_batch > 0: this plan is one slab of a batch of that many transforms (mifft_plan_create_slab)
        self.whole_batch = int(whole_batch)
        check(_lib.lib().mifft_plan_create_slab(ctypes.byref(h), self.device, _DTYPE_CODE[in_dtype],
                                                _DTYPE_CODE[out_dtype], len(dims), c_dims, out_shape[0], in_shape[-1],
                                                int(self.inverse), c_flat, c_len, self.flags, self.whole_batch))
        self._h = h

    # -- introspection ------------------------------------------------------
    @property
    def ndim(self) -> int:
        return len(self.out_shape) - 2

    def stages(self, dim: int) -> list:
        out = (ctypes.c_uint32 * 64)()
        k = check(_lib.lib().mifft_plan_stages(self._h, dim, out, 64))
        return list(out[:k])

    def kernel_name(self, dim: int) -> str:
        return _lib.lib().mifft_plan_kernel_name(self._h, dim).decode()

    @property
    def num_launches(self) -> int:
        return check(_lib.lib().mifft_plan_num_launches(self._h))

    @property
    def in_bytes(self) -> int:
        return int(_lib.lib().mifft_plan_in_bytes(self._h))

    @property
    def out_bytes(self) -> int:
        return int(_lib.lib().mifft_plan_out_bytes(self._h))

    @property
    def scratch_bytes(self) -> int:
        """device bytes of the plan-owned scratch tensor (0 unless a long strided dimension or a three-launch
        four-step needs one; the reference's plan always owns one, fft/fft/_ndim_fft_gpu.mojo:185)"""
        return int(_lib.lib().mifft_plan_scratch_bytes(self._h))

    def device_status(self, ctx: Optional["DeviceContext"] = None) -> int:
        """Device-side error flags raised by execs of this plan (read and cleared; 0 = none).  Only the opt-in
        L2-resident image kernel can raise one (bounded XCD-barrier spin expired: bit 0; surplus workgroup: bit 1);
        synchronises the stream in that case only."""
        flags = ctypes.c_uint32(0)
        stream = (ctx.stream if ctx is not None else torch.cuda.current_stream(self.device)).cuda_stream
        check(_lib.lib().mifft_plan_device_status(self._h, stream, ctypes.byref(flags)))
        return int(flags.value)

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().mifft_plan_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_fft(in_dtype, out_dtype, in_layout: Sequence[int], out_layout: Sequence[int], *, bases=None,
             inverse: bool = False, runtime_twfs: bool = True, max_cluster_size: int = 8,
             _test: Optional[GPUTest] = None, faithful_stages: bool = False,
             ctx: Optional[DeviceContext] = None, whole_batch: int = 0) -> Plan:
    """GPU overload of plan_fft (fft/fft/fft.mojo:161-210).

    ``runtime_twfs`` and ``max_cluster_size`` are accepted for call-site compatibility
    and ignored: twiddles always come from an fp64-accurate device table and CDNA4
    has no thread-block clusters.  ``bases=None`` selects the reference's GPU default.
    ``whole_batch`` (no reference counterpart): this plan covers one slab of a batch of that many transforms split over
    several plans / GPUs; size-dependent kernel choices follow the whole batch, so the slab's results equal the same
    rows of one plan over the whole batch bit for bit.
    """
    del runtime_twfs, max_cluster_size
    if ctx is None:
        ctx = DeviceContext()
    flags = FLAG_FAITHFUL_STAGES if (faithful_stages or _test is not None) else 0
    return Plan(in_dtype, out_dtype, in_layout, out_layout, bases=bases, inverse=inverse,
                device=ctx.device, flags=flags, whole_batch=whole_batch)


def _check_tensor(t: "torch.Tensor", shape: tuple, dtype, device: int, what: str) -> None:
    if not t.is_cuda or t.device.index != device:
        raise MifftError(-10, f"{what} must live on HIP device {device}, got {t.device}")
    if t.dtype != dtype or tuple(t.shape) != shape:
        raise MifftError(-14, f"{what} is {tuple(t.shape)} {t.dtype}, plan expects {shape} {dtype}")
    if not t.is_contiguous():
        raise MifftError(-14, f"{what} must be row-major contiguous")


def fft(output: "torch.Tensor", x: "torch.Tensor", ctx: Optional[DeviceContext] = None, *, plan: Plan,
        first: int = 0, count: Optional[int] = None) -> None:
    """GPU overload of fft (fft/fft/fft.mojo:262-323): enqueue on ctx's stream and return.

    Out of place; ``x`` is never written; every element of ``output`` (of the selected
    batch range) is written.  ``first``/``count`` select a slab of the leading dimension
    (used by the batch-sharded multi-GPU host); default is the whole batch.
    """
    if ctx is None:
        ctx = DeviceContext(plan.device)
    _check_tensor(x, plan.in_shape, plan.in_dtype, plan.device, "x")
    _check_tensor(output, plan.out_shape, plan.out_dtype, plan.device, "output")
    if count is None:
        count = plan.out_shape[0] - first
    check(_lib.lib().mifft_exec_batch(plan._h, x.data_ptr(), output.data_ptr(), int(first), int(count),
                                      ctx.stream.cuda_stream))


def time_fft(output: "torch.Tensor", x: "torch.Tensor", *, plan: Plan, iters: int = 10,
             ctx: Optional[DeviceContext] = None) -> float:
    """Average milliseconds per exec over ``iters`` back-to-back execs, measured with HIP
    events recorded on the launch stream inside the library (mifft_time_exec)."""
    if ctx is None:
        ctx = DeviceContext(plan.device)
    _check_tensor(x, plan.in_shape, plan.in_dtype, plan.device, "x")
    _check_tensor(output, plan.out_shape, plan.out_dtype, plan.device, "output")
    ms = ctypes.c_float()
    check(_lib.lib().mifft_time_exec(plan._h, x.data_ptr(), output.data_ptr(), ctx.stream.cuda_stream,
                                     int(iters), ctypes.byref(ms)))
    return float(ms.value)


# ---------------------------------------------------------------------------
# convenience wrappers: fftn / ifftn / rfftn(shape, radices) call surface
# ---------------------------------------------------------------------------

def _as_interleaved(x: "torch.Tensor"):
    """complex (batch, d0..) -> real view (batch, d0.., 2); real-typed input must already be (batch, d0.., C)."""
    if x.is_complex():
        return torch.view_as_real(x.contiguous()), True
    return x.contiguous(), False


_PLAN_CACHE: "collections.OrderedDict" = collections.OrderedDict()
_PLAN_CACHE_SIZE = 32
_PLAN_CACHE_SCRATCH_BYTES = 8 << 30  # ... and at most this much plan-owned scratch (long-strided / three-launch routes)
_PLAN_CACHE_LOCK = threading.RLock()


def _cached_plan(in_dtype, out_dtype, in_shape, out_shape, radices, inverse, faithful_stages, device) -> Plan:
    """Plans of the convenience wrappers are kept (LRU): a plan is a few small device tables, building one
    costs a hipMalloc + copy per dimension, and its tables must outlive the kernels enqueued with it.

    A plan may own mutable device state that its execs share (the scratch tensor of the long-strided and three-launch
    four-step routes, the counters of the opt-in image kernel), so the C ABI allows ONE exec in flight per plan
    (include/mifft.h).  Execs are ordered on a stream; the cache therefore keeps one plan PER STREAM -- two streams (or
    two threads on their own streams) transforming the same shape never share a plan -- and is guarded by a lock."""
    with _PLAN_CACHE_LOCK:
        return _cached_plan_locked(in_dtype, out_dtype, in_shape, out_shape, radices, inverse, faithful_stages, device)


def _cached_plan_locked(in_dtype, out_dtype, in_shape, out_shape, radices, inverse, faithful_stages, device) -> Plan:
    key = (in_dtype, out_dtype, in_shape, out_shape,
           None if radices is None else tuple(tuple(int(b) for b in r) for r in radices),
           bool(inverse), bool(faithful_stages), device, int(torch.cuda.current_stream(device).cuda_stream))
    plan = _PLAN_CACHE.get(key)
    if plan is None:
        try:
            plan = plan_fft(in_dtype, out_dtype, in_shape, out_shape, bases=radices, inverse=inverse,
                            faithful_stages=faithful_stages, ctx=DeviceContext(device))
        except MifftError as e:
            # plan_fft keeps the reference's behaviour: its default radix estimate (trial division by 2..32 on the GPU,
            # primes <= 97 otherwise, fft/fft/fft.mojo:49-104) rejects lengths with a larger prime factor.  The
            # numpy-style wrappers are this repository's own surface, so they retry with the full prime factorisation.
            if radices is not None or e.status not in (-5, -7):
                raise
            plan = plan_fft(in_dtype, out_dtype, in_shape, out_shape, bases=[_prime_factors(int(n)) for n in in_shape[1:-1]],
                            inverse=inverse, faithful_stages=faithful_stages, ctx=DeviceContext(device))
        _PLAN_CACHE[key] = plan
        while len(_PLAN_CACHE) > 1 and (len(_PLAN_CACHE) > _PLAN_CACHE_SIZE or
                                        sum(q.scratch_bytes for q in _PLAN_CACHE.values()) > _PLAN_CACHE_SCRATCH_BYTES):
            _, old = _PLAN_CACHE.popitem(last=False)
            torch.cuda.synchronize(old.device)  # nothing enqueued with the evicted plan may still run
            old.close()
    else:
        _PLAN_CACHE.move_to_end(key)
    return plan


def _prime_factors(n: int) -> list:
    """distinct prime factors of n, ascending (a complete `bases` list for any length)"""
    f, d = [], 2
    while d * d <= n:
        if n % d == 0:
            f.append(d)
            while n % d == 0:
                n //= d
        d += 1
    if n > 1:
        f.append(n)
    return f


def clear_plan_cache() -> None:
    with _PLAN_CACHE_LOCK:
        for plan in _PLAN_CACHE.values():
            torch.cuda.synchronize(plan.device)
            plan.close()
        _PLAN_CACHE.clear()


def _run(x: "torch.Tensor", *, radices, inverse: bool, out_dtype, faithful_stages: bool) -> "torch.Tensor":
    xr, was_complex = _as_interleaved(x)
    if out_dtype is None:
        out_dtype = xr.dtype if xr.dtype in (torch.float32, torch.float64) else torch.float64
    out_shape = tuple(xr.shape[:-1]) + (2,)
    out = torch.empty(out_shape, dtype=out_dtype, device=xr.device)
    # lookup AND enqueue under the cache lock: another thread that inserts a plan may evict (synchronise + close) only
    # plans that have nothing left to enqueue
    with _PLAN_CACHE_LOCK:
        plan = _cached_plan_locked(xr.dtype, out_dtype, tuple(xr.shape), out_shape, radices, inverse, faithful_stages,
                                   xr.device.index)
        fft(out, xr, DeviceContext(xr.device.index), plan=plan)  # asynchronous on the current stream, like torch ops
    return torch.view_as_complex(out) if was_complex else out


def fftn(x: "torch.Tensor", radices=None, *, out_dtype=None, faithful_stages: bool = False) -> "torch.Tensor":
    """Forward C2C transform over every dim but the first (batch).  ``x``: complex
    ``(batch, d0..)`` or real-typed interleaved ``(batch, d0.., 2)``; ``radices``: one list per dim.
    Asynchronous on the current torch stream; plans are cached per (shape, dtype, radices, device, stream), so
    concurrent streams never share a plan's scratch (one exec in flight per plan, include/mifft.h)."""
    return _run(x, radices=radices, inverse=False, out_dtype=out_dtype, faithful_stages=faithful_stages)


def ifftn(x: "torch.Tensor", radices=None, *, out_dtype=None, faithful_stages: bool = False) -> "torch.Tensor":
    """Inverse C2C transform (1/N per dimension), same layout rules as fftn."""
    return _run(x, radices=radices, inverse=True, out_dtype=out_dtype, faithful_stages=faithful_stages)


def rfftn(x: "torch.Tensor", radices=None, *, out_dtype=None, faithful_stages: bool = False) -> "torch.Tensor":
    """Real-input transform: ``x`` is real ``(batch, d0..)``; returns the FULL spectrum as
    interleaved ``(batch, d0.., 2)`` like the reference (fft/fft/_fft.mojo:254-257), not numpy's half spectrum."""
    if x.is_complex():
        raise MifftError(-3, "rfftn expects a real tensor")
    return _run(x.unsqueeze(-1), radices=radices, inverse=False, out_dtype=out_dtype,
                faithful_stages=faithful_stages)
