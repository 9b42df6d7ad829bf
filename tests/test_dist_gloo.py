"""Multi-process batch sharding over gloo on the CPU (world_size 2 and 3).

The HIP library cannot run here, so the local transform injected into ShardedFFT is the CPU
oracle; what is under test is the product's sharding arithmetic and its P2P scatter / gather
plumbing.  The GPU path of the same class is exercised by bench.py --gpus N on the driver's node.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from hackathon_fft_amd.dist import ShardedFFT, all_shard_bounds, shard_bounds


def test_shard_bounds_are_a_balanced_partition():
    for batch in (0, 1, 7, 10, 100000):
        for world in (1, 2, 3, 4, 8):
            b = all_shard_bounds(batch, world)
            assert b[0][0] == 0 and sum(c for _, c in b) == batch
            for (f0, c0), (f1, _) in zip(b, b[1:]):
                assert f0 + c0 == f1
            counts = [c for _, c in b]
            assert max(counts) - min(counts) <= 1
    assert all_shard_bounds(10, 8) == [(0, 2), (2, 2), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1), (9, 1)]
    assert shard_bounds(100000, 8, 3) == (37500, 12500)
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


class _OracleBackend:
    def __init__(self, in_dtype, out_dtype, in_shape, out_shape, bases, inverse):
        from oracle import mifft_oracle as O
        self.O = O
        tn = {torch.float32: np.float32, torch.float64: np.float64}
        self.plan = O.plan_fft(tn[in_dtype], tn[out_dtype], in_shape, out_shape, bases=bases, inverse=inverse,
                               default_target="gpu")

    def run(self, out, x, first=0, count=None):
        if count is None:
            count = out.shape[0] - first
        o = np.full(tuple(out.shape), np.nan, dtype=out.numpy().dtype)
        self.O.fft(o, np.ascontiguousarray(x.numpy()), plan=self.plan, cpu_workers=1, first=first, count=count)
        out[first:first + count].copy_(torch.from_numpy(o[first:first + count]))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, shape, bases, q, loopback=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = tuple(shape) + (2,)
        sh = ShardedFFT(torch.float32, torch.float32, full, full, bases=bases, local_backend=_OracleBackend)
        # --- resident shards, no communication ---
        rng = np.random.default_rng(1234)
        x_all = torch.from_numpy(rng.standard_normal(full).astype(np.float32))  # same on every rank (same seed)
        x_slab = x_all[sh.first:sh.first + sh.count].contiguous()
        out_slab = torch.full(sh.slab_out_shape, float("nan"))
        sh.fft(out_slab, x_slab)
        gathered = [None] * world
        dist.all_gather_object(gathered, out_slab.numpy())
        # --- root-held tensor: scatter, transform, gather ---
        out_full = torch.full(full, float("nan")) if rank == 0 else None
        sh.fft_from_root(out_full, x_all if rank == 0 else None, root=0, device=torch.device("cpu"),
                         loopback=loopback)
        if rank == 0:
            from oracle import mifft_oracle as O
            ref = O.fftn(x_all.numpy(), bases=bases)
            resident = np.concatenate([g for g in gathered if g.shape[0]], axis=0)
            q.put((np.array_equal(resident, ref), np.array_equal(out_full.numpy(), ref),
                   bool(np.isnan(out_full.numpy()).any())))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,bases,loopback", [
    (2, (10, 64), [[2]], False),
    (2, (5, 12, 10), None, False),
    (3, (10, 93), [[31, 3]], False),   # uneven 4,3,3
    (3, (2, 16), None, False),         # a rank with an empty slab
    (2, (37, 32), None, False),        # 19 / 18 entries per rank: eight chunks of 3 / 2 entries in the pipeline
    (3, (50, 20), None, False),        # 17 / 17 / 16
    (1, (3, 64), [[2]], True),         # one rank, its slab sent to itself through the process group
    (2, (5, 12, 10), None, True),      # the root's own slab takes the P2P path too
])
def test_sharded_equals_single_rank_bit_for_bit(world, shape, bases, loopback):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, bases, q, loopback)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        resident_ok, root_ok, has_nan = q.get(timeout=120)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert resident_ok, "concatenated shard outputs must equal the single-rank output bit for bit"
    assert root_ok and not has_nan
    assert all(p.exitcode == 0 for p in procs)


def test_bench_launcher_starts_one_process_per_rank():
    """`python bench.py --gpus 2` with no WORLD_SIZE must itself start two rank processes (before torch or HIP is
    touched).  There is no GPU here, so each rank stops at the device check -- which is enough to see that two ranks
    with WORLD_SIZE=2 were started and that the launcher reports their failure."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu rehearsal test")
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a HIP device") == 2, r.stderr[-2000:]
    assert "rank exit codes" in r.stderr and r.stdout.strip() == ""


def test_bench_rejects_a_world_size_that_differs_from_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
