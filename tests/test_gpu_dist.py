"""The multi-GPU host path on real hardware, as far as ONE GPU allows.

* ShardedFFT.fft_from_root over the `nccl` backend (= RCCL) at world_size 1 with `loopback=True`: the root's slab is
  sent to itself through a grouped RCCL send/recv pair, transformed by libmifft, and gathered back the same way -- the
  collective code path runs on the GPU and must reproduce the direct result bit for bit.
* `python bench.py --gpus 2` with no launcher: bench.py itself starts the two ranks (fresh processes, before anything
  touches HIP).  RCCL refuses two ranks on one device, so the rehearsal flag puts both ranks on the visible GPU and
  moves the (host-staged) slabs over gloo; what is checked is the launch, the split, ranks_seen and the JSON contract.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_RCCL_LOOPBACK = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
import hackathon_fft_amd as mf
from hackathon_fft_amd.dist import ShardedFFT
from oracle import mifft_oracle as O

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r, RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    for shape, bases in [((6, 1024), [[2]]), ((3, 16, 16, 16), None), ((5, 93), [[31, 3]])]:
        full = tuple(shape) + (2,)
        rng = np.random.default_rng(99)
        xh = rng.standard_normal(full).astype(np.float32)
        x = torch.from_numpy(xh).cuda()
        sh = ShardedFFT(torch.float32, torch.float32, full, full, bases=bases, device=0)
        assert sh.world == 1 and dist.get_backend() == "nccl"
        direct = torch.full_like(x, float("nan"))
        sh.fft(direct, x)
        looped = torch.full_like(x, float("nan"))
        sh.fft_from_root(looped, x, root=0, loopback=True)   # RCCL send-to-self / recv-from-self, both directions
        torch.cuda.synchronize()
        assert torch.equal(direct, looped), shape
        ref = O.fftn(xh, bases=bases).astype(np.float64)
        got = looped.cpu().numpy().astype(np.float64)
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert err < 1e-5 and not np.isnan(got).any(), (shape, err)
    print("RCCL_LOOPBACK_OK")
finally:
    dist.destroy_process_group()
"""


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_fft_from_root_runs_over_rccl_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_LOOPBACK % {"root": ROOT, "port": _free_port()}],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_LOOPBACK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu",
                        "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["config"]["workload"] == "1d_100kx1024_radix2" and d["steps"] == 5 and d["warmup"] == 2
    s = d["strong_config5"]
    assert s["volumes_per_rank"] == [5, 5] and s["ideal_speedup_vs_1gpu"] == 2.0
    assert s["compute_shards_resident"]["ms_per_step"] > 0 and s["end_to_end_from_root"]["ms_per_step"] > 0


def test_bench_single_gpu_line_carries_every_baseline_config():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3",
                        "--no-cpu-baseline", "--no-compare-vendor"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["roofline"]["bound"] == "hbm" and d["ramp"]["execs"] > 0
    # HBM traffic from PMC counters collected by this very run (rocprofv3 child passes), within 1 % of the algorithmic bytes
    assert d["roofline"]["traffic_source"].startswith("live")
    assert abs(d["roofline"]["traffic"] / d["roofline"]["algorithmic_bytes_per_launch"] - 1.0) < 0.01
    assert [c["baseline_config_index"] for c in d["configs"]] == [0, 2, 3, 4]
    for c in d["configs"]:
        assert 0 < c["roofline"]["frac"] < 1 and c["ms_per_step"] > 0 and c["kernels"]
    assert d["strong_config5"]["volumes_per_rank"] == [10]
    # the reference's own benchmark mode (real input) on its own shapes
    assert len(d["rfft_reference_bench"]) == 7
    for c in d["rfft_reference_bench"]:
        assert c["shape"][-1] == 1 and 0 < c["roofline_frac"] < 1 and "generic" not in c["kernels"]


@pytest.mark.parametrize("shape,world", [((100000, 1024), 8), ((250000, 128), 4), ((30000, 1024), 3), ((10, 128, 128, 128), 8)])
def test_slab_plans_choose_their_kernels_for_the_whole_batch(shape, world):
    """mifft_plan_create_slab: size-dependent choices (streaming / non-temporal twins, cache policy) follow the WHOLE
    batch, so every rank's slab equals the same rows of the single-GPU result bit for bit."""
    import torch
    import hackathon_fft_amd as mf
    from hackathon_fft_amd.dist import all_shard_bounds
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(8)
    x = torch.randn(shape + (2,), generator=g, device=dev)
    ctx = mf.DeviceContext(0)
    whole = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
    out = torch.empty_like(x)
    mf.fft(out, x, ctx, plan=whole)
    names = [whole.kernel_name(d) for d in range(len(shape) - 1)]
    for first, count in all_shard_bounds(shape[0], world):
        xs = x[first:first + count].contiguous()
        slab = mf.plan_fft(torch.float32, torch.float32, xs.shape, xs.shape, ctx=ctx, whole_batch=shape[0])
        assert [slab.kernel_name(d) for d in range(len(shape) - 1)] == names
        os_ = torch.full_like(xs, float("nan"))
        mf.fft(os_, xs, ctx, plan=slab)
        ctx.synchronize()
        assert torch.equal(os_, out[first:first + count]), (shape, first, count)


@pytest.mark.parametrize("shape,world", [((10, 128, 128, 128), 8), ((100, 640, 480), 4), ((7, 256, 256, 256), 3)])
def test_slab_plans_of_real_input_keep_the_hermitian_schedule(shape, world):
    """Real input: the whole batch takes a Hermitian last pass behind a half-store pass (herm_pays counts the rounds of the
    persistent grid for the WHOLE batch); a slab plan must take the same kernels -- a slab by itself would often be too small
    for them -- and equal the same rows bit for bit."""
    import torch
    import hackathon_fft_amd as mf
    from hackathon_fft_amd.dist import all_shard_bounds
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(shape + (1,), generator=g, device=dev)
    out_shape = shape + (2,)
    ctx = mf.DeviceContext(0)
    whole = mf.plan_fft(torch.float32, torch.float32, x.shape, out_shape, ctx=ctx)
    out = torch.empty(out_shape, device=dev)
    mf.fft(out, x, ctx, plan=whole)
    names = [whole.kernel_name(d) for d in range(len(shape) - 1)]
    assert names[0].endswith(("_h", "_h_jit")) and any("_hs" in n for n in names[1:]), names
    for first, count in all_shard_bounds(shape[0], world):
        xs = x[first:first + count].contiguous()
        slab = mf.plan_fft(torch.float32, torch.float32, xs.shape, (count,) + out_shape[1:], ctx=ctx, whole_batch=shape[0])
        assert [slab.kernel_name(d) for d in range(len(shape) - 1)] == names
        os_ = torch.full((count,) + out_shape[1:], float("nan"), device=dev)
        mf.fft(os_, xs, ctx, plan=slab)
        ctx.synchronize()
        assert torch.equal(os_, out[first:first + count]), (shape, first, count)
