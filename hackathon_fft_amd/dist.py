"""Batch-sharded multi-GPU execution: one process per GPU over torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm).

The reference has no multi-device code; its leading dimension is pure batch
(fft/fft/_ndim_fft_gpu.mojo:61-63,428-450; fft/fft/_ndim_fft_cpu.mojo:120,323), so every
batch entry is an independent transform and the path shards with NO exchange during compute:

  * resident shards (``ShardedFFT.fft``): each rank owns a contiguous slab of the leading
    dimension and runs the single-GPU plan on it -- zero communication.  This is what
    ``bench.py --gpus N`` measures (weak scaling).
  * root-held tensor (``ShardedFFT.fft_from_root``): the only collective step is the batch
    split itself -- the root posts one send per peer and every peer one receive (grouped P2P,
    so all 7 xGMI links of the root carry traffic concurrently instead of a ring being bound
    by one link), slabs are transformed locally, and the results return the same way.

Slabs are balanced: ``b_r = B // G + (r < B % G)``.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world_size: int, rank: int) -> Tuple[int, int]:
    """(first, count) of rank's contiguous slab of the leading dimension."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank / world_size")
    q, r = divmod(int(batch), world_size)
    count = q + (1 if rank < r else 0)
    first = rank * q + min(rank, r)
    return first, count


def all_shard_bounds(batch: int, world_size: int) -> List[Tuple[int, int]]:
    return [shard_bounds(batch, world_size, r) for r in range(world_size)]


class _HipBackend:
    """Local transform through libmifft (the product path)."""

    def __init__(self, in_dtype, out_dtype, in_shape, out_shape, bases, inverse, device, whole_batch=0):
        from . import api
        self._api = api
        self.ctx = api.DeviceContext(device)
        # size-dependent kernel choices follow the WHOLE batch: concatenated slabs == the single-GPU result, bit for bit
        self.plan = api.plan_fft(in_dtype, out_dtype, in_shape, out_shape, bases=bases, inverse=inverse, ctx=self.ctx,
                                 whole_batch=whole_batch)

    def run(self, out: torch.Tensor, x: torch.Tensor) -> None:
        self._api.fft(out, x, self.ctx, plan=self.plan)


class ShardedFFT:
    """One plan per rank for its slab of a ``(batch, d0.., C)`` problem.

    ``local_backend`` is a factory ``(in_dtype, out_dtype, in_shape, out_shape, bases, inverse)
    -> object with .run(out, x)``; the default runs libmifft on this rank's GPU.  ``match_single_gpu`` (default): the
    slab plans make their size-dependent kernel choices for the WHOLE batch (mifft_plan_create_slab), so the concatenated
    slabs equal the result of one plan on one GPU bit for bit; False lets every rank choose for its own slab size.  (The CPU test-suite
    injects the oracle here to exercise the sharding and the P2P plumbing over gloo.)
    """

    def __init__(self, in_dtype, out_dtype, in_shape: Sequence[int], out_shape: Sequence[int], *, bases=None,
                 inverse: bool = False, group=None, device: Optional[int] = None,
                 local_backend: Optional[Callable] = None, match_single_gpu: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.in_shape, self.out_shape = tuple(in_shape), tuple(out_shape)
        self.in_dtype, self.out_dtype = in_dtype, out_dtype
        self.batch = self.in_shape[0]
        self.bounds = all_shard_bounds(self.batch, self.world)
        self.first, self.count = self.bounds[self.rank]
        self.slab_in_shape = (self.count,) + self.in_shape[1:]
        self.slab_out_shape = (self.count,) + self.out_shape[1:]
        self._backend = None
        if self.count > 0:
            if local_backend is None:
                if device is None:
                    device = torch.cuda.current_device()
                self._backend = _HipBackend(in_dtype, out_dtype, self.slab_in_shape, self.slab_out_shape, bases,
                                            inverse, device, whole_batch=self.batch if match_single_gpu else 0)
            else:
                self._backend = local_backend(in_dtype, out_dtype, self.slab_in_shape, self.slab_out_shape, bases,
                                              inverse)

    # ---- resident shards: no communication ----------------------------------------------
    def fft(self, out_slab: torch.Tensor, x_slab: torch.Tensor) -> None:
        if tuple(x_slab.shape) != self.slab_in_shape or tuple(out_slab.shape) != self.slab_out_shape:
            raise ValueError(f"rank {self.rank} expects slabs {self.slab_in_shape} -> {self.slab_out_shape}")
        if self.count:
            self._backend.run(out_slab, x_slab)

    # ---- root-held tensor: split, transform, collect -------------------------------------
    def _p2p(self, transfers):
        """``transfers``: list of ("send" | "recv", tensor, peer).  One grouped batch (ncclGroupStart/End under RCCL, so
        the root's sends to all peers -- and a loopback pair -- progress concurrently).  A process group that cannot move
        device memory (gloo) gets pinned host staging copies; the transform itself always runs on the GPU."""
        if not transfers:
            return
        backend = dist.get_backend(self.group)
        if backend == "gloo":
            # gloo has no pair to itself: a loopback transfer is a local copy there (RCCL executes it as a real
            # grouped send/recv, which is what the loopback mode exists for)
            me = [tr for tr in transfers if tr[2] == self.rank]
            sends, recvs = [t for k, t, _ in me if k == "send"], [t for k, t, _ in me if k == "recv"]
            for src, dst in zip(sends, recvs):
                dst.copy_(src)
            transfers = [tr for tr in transfers if tr[2] != self.rank]
        staged = []
        ops = []
        for kind, t, peer in transfers:
            buf = t
            if backend == "gloo" and t.is_cuda:
                buf = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
                if kind == "send":
                    buf.copy_(t)
                else:
                    staged.append((t, buf))
            elif not t.is_contiguous():
                raise ValueError("slabs of the leading dimension are contiguous by construction")
            ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, buf, peer, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for t, buf in staged:
            t.copy_(buf)

    def fft_from_root(self, out_full: Optional[torch.Tensor], x_full: Optional[torch.Tensor], *, root: int = 0,
                      device=None, loopback: bool = False) -> None:
        """``x_full`` / ``out_full`` are read / written on ``root`` only (may be None elsewhere).

        ``loopback=True`` also moves the ROOT's own slab through the process group (a grouped send-to-self /
        receive-from-self pair into staging slabs) instead of transforming it where it lies.  It exists so that the
        collective code path can be executed and checked on a single GPU (world_size 1 over RCCL); it costs two
        extra device copies and is never what a production call wants."""
        if self.world == 1 and not loopback:
            self.fft(out_full, x_full)
            return
        is_root = self.rank == root
        if not is_root or loopback:
            dev = x_full.device if is_root else (
                device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                                   if torch.cuda.is_available() else torch.device("cpu")))
            x_slab = torch.empty(self.slab_in_shape, dtype=self.in_dtype, device=dev)
            out_slab = torch.empty(self.slab_out_shape, dtype=self.out_dtype, device=dev)
        else:
            x_slab = x_full[self.first:self.first + self.count]
            out_slab = out_full[self.first:self.first + self.count]
        tr = []
        if is_root:
            tr += [("send", x_full[f:f + c], r) for r, (f, c) in enumerate(self.bounds)
                   if c > 0 and (r != root or loopback)]
        if self.count > 0 and (not is_root or loopback):
            tr.append(("recv", x_slab, root))
        self._p2p(tr)
        if self.count:
            self._backend.run(out_slab, x_slab)
            if out_slab.is_cuda:
                torch.cuda.current_stream(out_slab.device).synchronize()
        tr = []
        if self.count > 0 and (not is_root or loopback):
            tr.append(("send", out_slab, root))
        if is_root:
            tr += [("recv", out_full[f:f + c], r) for r, (f, c) in enumerate(self.bounds)
                   if c > 0 and (r != root or loopback)]
        self._p2p(tr)
