"""Batch-sharded multi-GPU execution: one process per GPU over torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm).

The reference has no multi-device code; its leading dimension is pure batch
(fft/fft/_ndim_fft_gpu.mojo:61-63,428-450; fft/fft/_ndim_fft_cpu.mojo:120,323), so every
batch entry is an independent transform and the path shards with NO exchange during compute:

  * resident shards (``ShardedFFT.fft``): each rank owns a contiguous slab of the leading
    dimension and runs the single-GPU plan on it -- zero communication.  This is what
    ``bench.py --gpus N`` measures (weak scaling).
  * root-held tensor (``ShardedFFT.fft_from_root``): the only collective step is the batch
    split itself, run as a PIPELINE of per-entry chunks: the root posts the sends of every
    chunk to every peer up front (grouped P2P, so all 7 xGMI links of the root carry traffic
    concurrently instead of a ring being bound by one link) and the receives of every result
    chunk on a SECOND process group (its own communicator and stream: results flow back while
    later chunks still go out -- xGMI links are full duplex); a peer posts all its receives,
    transforms chunk k as soon as chunk k has arrived (stream-ordered waits under RCCL, no host
    synchronisation anywhere) and sends its result back at once.

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

    def run(self, out: torch.Tensor, x: torch.Tensor, first: int = 0, count: Optional[int] = None) -> None:
        """entries [first, first + count) of the slab (default: all of it); bit-identical to the whole-slab exec"""
        self._api.fft(out, x, self.ctx, plan=self.plan, first=first, count=count)


class ShardedFFT:
    """One plan per rank for its slab of a ``(batch, d0.., C)`` problem.

    ``local_backend`` is a factory ``(in_dtype, out_dtype, in_shape, out_shape, bases, inverse)
    -> object with .run(out, x, first=0, count=None)``; the default runs libmifft on this rank's GPU.
    ``match_single_gpu`` (default): the slab plans make their size-dependent kernel choices for the WHOLE batch
    (mifft_plan_create_slab), so the concatenated slabs equal the result of one plan on one GPU bit for bit.  The price:
    a slab that is cache-resident on its own (10 x 128^3 over 8 ranks: 17-34 MB per rank) still runs the kernels and the
    non-temporal / cache policy chosen for the whole 168-MB batch, which this library's own probes put 6-9 % behind the
    small-tensor choice.  ``match_single_gpu=False`` lets every rank choose for its own slab size
    (results then agree with the single-GPU plan to rounding, not bit for bit); bench.py's strong-scaling leg times both.
    (The CPU test-suite injects the oracle as ``local_backend`` to exercise the sharding and the P2P plumbing over gloo.)

    ``fft_from_root`` is a COLLECTIVE call: every rank of ``group`` calls it at the same point (the first call also
    creates the second process group the pipeline returns its results on).
    """

    def __init__(self, in_dtype, out_dtype, in_shape: Sequence[int], out_shape: Sequence[int], *, bases=None,
                 inverse: bool = False, group=None, device: Optional[int] = None,
                 local_backend: Optional[Callable] = None, match_single_gpu: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # results return on their own group = their own communicator (and, under RCCL, their own stream): the gather of
        # chunk k overlaps the scatter of chunk k + 1.  Created by the first fft_from_root (a collective call anyway), so
        # that resident-shard use never creates a second communicator.
        self.group_back = None
        self.max_chunks = 8  # chunks per slab in fft_from_root (one chunk >= one batch entry)
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
    def _chunks(self, count: int) -> List[Tuple[int, int]]:
        """(first, count) pieces of a slab of `count` entries: at most max_chunks, whole batch entries"""
        if count <= 0:
            return []
        n = min(self.max_chunks, count)
        q, r = divmod(count, n)
        out, f = [], 0
        for i in range(n):
            c = q + (1 if i < r else 0)
            out.append((f, c))
            f += c
        return out

    def _post(self, kind: str, pieces, group, staged=None):
        """One grouped batch of sends or receives (``pieces``: [(tensor, peer)]); returns the work handles.  A process group
        that cannot move device memory (gloo: the 1-GPU rehearsal and the CPU tests) gets pinned host staging buffers."""
        if not pieces:
            return []
        gloo = dist.get_backend(group) == "gloo"
        ops = []
        for t, peer in pieces:
            if not t.is_contiguous():
                raise ValueError("slabs of the leading dimension are contiguous by construction")
            buf = t
            if gloo and t.is_cuda:
                buf = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
                if kind == "send":
                    torch.cuda.current_stream(t.device).synchronize()  # (rehearsal path only)
                    buf.copy_(t)
                else:
                    staged.append((t, buf))
            ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, buf, peer, group))
        return dist.batch_isend_irecv(ops)

    @staticmethod
    def _wait(works, staged=None):
        """Under RCCL `wait()` orders the CURRENT STREAM behind the transfer and returns at once; under gloo it blocks."""
        for w in works:
            w.wait()
        for t, buf in (staged or []):
            t.copy_(buf)

    def fft_from_root(self, out_full: Optional[torch.Tensor], x_full: Optional[torch.Tensor], *, root: int = 0,
                      device=None, loopback: bool = False) -> None:
        """``x_full`` / ``out_full`` are read / written on ``root`` only (may be None elsewhere).

        Pipeline (module docstring): inputs go out and results come back in chunks of whole batch entries on two process
        groups; a peer transforms chunk k as soon as it has arrived.  Nothing in here synchronises the host with the
        device under RCCL -- the caller's stream is ordered behind the last transfer when the call returns, like any other
        stream-ordered operation; synchronise the stream before reading ``out_full`` on the host.

        ``loopback=True`` also moves the ROOT's own slab through the process group (a grouped send-to-self /
        receive-from-self pair into staging slabs) instead of transforming it where it lies.  It exists so that the
        collective code path can be executed and checked on a single GPU (world_size 1 over RCCL); it costs two
        extra device copies and is never what a production call wants.  Without an initialised process group the
        loopback degenerates to local copies."""
        if self.world == 1 and not loopback:
            self.fft(out_full, x_full)
            return
        is_root = self.rank == root
        if loopback and not dist.is_initialized():
            x_slab = x_full.clone()
            out_slab = torch.empty_like(out_full)
            self.fft(out_slab, x_slab)
            out_full.copy_(out_slab)
            return
        if self.group_back is None and self.world > 1:
            ranks = dist.get_process_group_ranks(self.group) if self.group is not None else list(range(dist.get_world_size()))
            self.group_back = dist.new_group(ranks=ranks, backend=dist.get_backend(self.group))
        back = self.group_back if self.group_back is not None else self.group
        gloo = dist.get_backend(self.group) == "gloo"
        if not is_root or loopback:
            dev = x_full.device if is_root else (
                device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                                   if torch.cuda.is_available() else torch.device("cpu")))
            x_slab = torch.empty(self.slab_in_shape, dtype=self.in_dtype, device=dev)
            out_slab = torch.empty(self.slab_out_shape, dtype=self.out_dtype, device=dev)
        else:
            x_slab = x_full[self.first:self.first + self.count]
            out_slab = out_full[self.first:self.first + self.count]

        pending, staged_all = [], []
        # ---- the root's own slab through the group (loopback): one grouped send + receive pair; gloo has no pair to
        #      itself, so the transfer is a local copy there ----
        if is_root and loopback and self.count:
            mine = x_full[self.first:self.first + self.count]
            if gloo:
                x_slab.copy_(mine)
            else:
                ops = [dist.P2POp(dist.isend, mine, root, self.group), dist.P2POp(dist.irecv, x_slab, root, self.group)]
                self._wait(dist.batch_isend_irecv(ops))
        # ---- 1. every input chunk goes out / every receive is posted, up front ----
        my_chunks = self._chunks(self.count)
        in_works = []   # peer: one entry per chunk
        if is_root:
            peers = [(r, f, self._chunks(c)) for r, (f, c) in enumerate(self.bounds) if r != root and c > 0]
            depth = max([len(ch) for _, _, ch in peers], default=0)
            for k in range(depth):  # chunk k of every peer in ONE group: the root's links work side by side
                pieces = [(x_full[f + ch[k][0]:f + ch[k][0] + ch[k][1]], r) for r, f, ch in peers if k < len(ch)]
                pending += self._post("send", pieces, self.group)
            # ... and the receives of every result chunk, on the second group
            for k in range(depth):
                st = []
                pieces = [(out_full[f + ch[k][0]:f + ch[k][0] + ch[k][1]], r) for r, f, ch in peers if k < len(ch)]
                pending += self._post("recv", pieces, back, st)
                staged_all += st
        elif self.count:
            for f, c in my_chunks:
                st = []
                in_works.append((self._post("recv", [(x_slab[f:f + c], root)], self.group, st), st))
        # ---- 2. transform: the root its own slab at once, a peer chunk by chunk as the chunks arrive ----
        if is_root:
            if self.count:
                self._backend.run(out_slab, x_slab)
        else:
            for (f, c), (works, st) in zip(my_chunks, in_works):
                self._wait(works, st)                       # stream-ordered under RCCL: no host synchronisation
                self._backend.run(out_slab, x_slab, first=f, count=c)
                pending += self._post("send", [(out_slab[f:f + c], root)], back)  # ordered behind the transform of chunk k
        # ---- 3. the root's loopback result, then completion of everything posted ----
        if is_root and loopback and self.count:
            dst = out_full[self.first:self.first + self.count]
            if gloo:
                dst.copy_(out_slab)
            else:
                ops = [dist.P2POp(dist.isend, out_slab, root, back), dist.P2POp(dist.irecv, dst, root, back)]
                self._wait(dist.batch_isend_irecv(ops))
        self._wait(pending, staged_all)
