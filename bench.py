#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A "step" is one pass of the hot path (one plan_fft'd `fft(out, x)` call) over one batch of
synthetic complex64 input that is already resident in HBM.  At N=1 the workload is
BASELINE.json configs[1]: 100k x 1024 1-D C2C fp32 with the user radix list [2]
("radix-2 Stockham").  For N>1 every rank owns its own 100k x 1024 slab (weak scaling, no
data-path collective: rows are independent transforms); value = rows of all ranks * FLOPs / the
max-over-ranks time.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      HBM roofline of the dominant kernel; `achieved` = algorithmic bytes per launch
                (16 B per complex element: one 8-B read + one 8-B write) divided by
                the average launch duration measured with HIP events on the launch stream inside
                libmifft (mifft_time_exec).
  cpu_baseline  the CPU oracle (restatement of the reference's multi-threaded CPU path) timed on
                this host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"

WORKLOADS = {
    # name: (shape without the complex dim, user bases, BASELINE.json config index)
    "1d_100kx1024_radix2": ((100000, 1024), [[2]], 1),
    "1d_500kx93_radix31x3": ((500000, 93), [[31, 3]], 2),
    "1d_500kx128": ((500000, 128), None, 0),
    "2d_100x640x480": ((100, 640, 480), None, 3),
    "3d_10x128x128x128": ((10, 128, 128, 128), None, 4),
    # further shapes of the reference's own bench / README tables (not BASELINE configs; index -1)
    "3d_100x64x64x64": ((100, 64, 64, 64), None, -1),
    "3d_1x256x256x256": ((1, 256, 256, 256), None, -1),
    "1d_100x16384": ((100, 16384), None, -1),
    "2d_10x1920x1080": ((10, 1920, 1080), None, -1),
    "2d_1x3840x2160": ((1, 3840, 2160), None, -1),
    "2d_1x7680x4320": ((1, 7680, 4320), None, -1),
    "1d_64x1048576_fourstep": ((64, 1 << 20), None, -1),
    # lengths without a precompiled kernel: the tile kernel specialised at plan creation (hipRTC)
    "1d_290kx343_radix7": ((290000, 343), None, -1),
    "1d_330kx97_prime": ((330000, 97), None, -1),
    "2d_3200x100x100_plane": ((3200, 100, 100), None, -1),
}
DEFAULT_WORKLOAD = "1d_100kx1024_radix2"


def flops_5nlogn(shape):
    n = 1
    for d in shape[1:]:
        n *= d
    return 5.0 * n * math.log2(n) * shape[0]


def host_cpu_share(nproc):
    """Cores this process may really use: min(affinity, cgroup cpu.max quota, nproc)."""
    n = nproc
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def measured_traffic(workload, kernels):
    """HBM bytes per launch from the committed PMC passes (profiles/rNN_<workload>.json, written by
    tools/summarize_prof.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of THIS command,
    FETCH_SIZE corrected by the calibrated gfx950 factor).  None when no profile of the current kernel
    set is committed -- counters cannot be collected from inside an un-profiled run."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{workload}.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        ks = [k for k in d.get("kernels", []) if "hbm_traffic_bytes" in k]
        if ks and len(ks) == len(set(kernels)):
            best = (sum(k["hbm_traffic_bytes"] for k in ks), os.path.basename(f))
    return best


def cpu_baseline(shape, bases, budget_s=12.0):
    """Oracle timed on a bounded sample (leading-batch subset) of the workload."""
    import numpy as np
    from oracle import mifft_oracle as O

    cores = host_cpu_share(O.num_procs())
    rng = np.random.default_rng(1234)
    per = int(np.prod(shape[1:]))

    def run(b):
        x = rng.standard_normal((b,) + tuple(shape[1:]) + (2,), dtype=np.float32)
        out = np.empty_like(x)
        plan = O.plan_fft(np.float32, np.float32, x.shape, x.shape, bases=bases, default_target="gpu")
        O.fft(out, x, plan=plan, cpu_workers=cores)  # warm-up (page faults, thread pool)
        t = time.perf_counter()
        O.fft(out, x, plan=plan, cpu_workers=cores)
        return time.perf_counter() - t

    b0 = max(1, min(shape[0], (1 << 21) // per))
    t0 = max(run(b0), 1e-4)
    b1 = int(max(b0, min(shape[0], b0 * (budget_s / 2) / t0, (1 << 28) // per)))
    t1 = run(b1) if b1 > b0 else t0
    gflops = flops_5nlogn((b1,) + tuple(shape[1:])) / t1 / 1e9
    return {
        "value": round(gflops, 3), "unit": "GFLOP/s", "cores": cores, "kind": "port",
        "sample": f"{b1} of {shape[0]} leading-batch entries of the same shape, all {cores} host threads, "
                  f"oracle/libmifft_oracle.so (C++ restatement of the reference CPU path), {t1 * 1e3:.1f} ms",
        "ms_per_transform": round(t1 * 1e3 / b1, 6),
    }


def _claim_stdout():
    """Keep stdout clean for the ONE JSON line: native libraries (RCCL prints a version banner on
    init) write to fd 1 directly, so fd 1 is pointed at stderr and the JSON goes to the saved fd."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(saved, "w")


def main():
    json_out = _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--faithful", action="store_true", help="force the literal stage-per-pass kernel family")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="arithmetic type (BASELINE metric: f32)")
    args = ap.parse_args()

    import torch

    import hackathon_fft_amd as mf

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MIFFT_BENCH_FORCE_DIST=1 runs the RCCL code path even with one rank (1-GPU rehearsal of the N>1 launch)
    distributed = world > 1 or os.environ.get("MIFFT_BENCH_FORCE_DIST") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: libmifft has no CPU path")
    torch.cuda.set_device(local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world

    shape, bases, cfg_idx = WORKLOADS[args.workload]
    dev = torch.device("cuda", local_rank)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    x = torch.randn(tuple(shape) + (2,), generator=gen, device=dev, dtype=tdt)
    out = torch.empty_like(x)
    ctx = mf.DeviceContext(local_rank)
    if distributed:
        # weak scaling: the global problem is world x (per-GPU shape); every rank owns one resident slab
        from hackathon_fft_amd.dist import ShardedFFT
        gshape = (shape[0] * world,) + tuple(shape[1:]) + (2,)
        sharded = ShardedFFT(tdt, tdt, gshape, gshape, bases=bases, device=local_rank)
        assert sharded.slab_in_shape == tuple(x.shape)
        plan, ctx = sharded._backend.plan, sharded._backend.ctx
    else:
        plan = mf.plan_fft(tdt, tdt, x.shape, x.shape, bases=bases, ctx=ctx, faithful_stages=args.faithful)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # clock / cache ramp: a short --warmup (a few ms of work) leaves the first timed steps 10-20 % slow, so the device
    # is kept busy for >= 50 ms before the W warmup steps (untimed, outside the K timed steps)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.05:
        for _ in range(10):
            mf.fft(out, x, ctx, plan=plan)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        mf.fft(out, x, ctx, plan=plan)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mf.fft(out, x, ctx, plan=plan)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel-level time: HIP events on the launch stream, inside the library
    launch_ms = mf.time_fft(out, x, plan=plan, iters=max(10, min(args.steps, 200)), ctx=ctx)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_flops = flops_5nlogn(shape) * n_gpus
        value = total_flops / (ms_per_step * 1e-3) / 1e9
        elems = 1
        for d in shape:
            elems *= d
        algo_bytes = 4.0 * esz * elems  # one complex read + one complex write per element, per exec on ONE gpu
        achieved = algo_bytes / (launch_ms * 1e-3) / 1e9
        kernels = [plan.kernel_name(d) for d in range(len(shape) - 1)]
        traffic = measured_traffic(args.workload, kernels)
        result = {
            "metric": "C2C GFLOP/s (5Nlog2N) + ms/transform, 100k×1024 fp32 @1/2/4/8 MI355X",
            "value": round(value, 2),
            "unit": "GFLOP/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5),
            "us_per_transform": round(ms_per_step * 1e3 / shape[0], 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "baseline_config_index": cfg_idx,
                "shape_per_gpu": list(shape) + [2],
                "bases": bases if bases is not None else "reference gpu default",
                "stages": [plan.stages(d) for d in range(len(shape) - 1)],
                "kernels": kernels,
                "launches_per_step": plan.num_launches,
                "parallelism": f"batch-sharded x{n_gpus}, no data-path collective",
                "input": "complex64 N(0,1), seed 1234+rank, resident in HBM",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic[0] if traffic else None,
                "traffic_source": traffic[1] if traffic else None,
                "kernel": "+".join(kernels),
                "algorithmic_bytes_per_launch": algo_bytes,
                "launch_ms_hip_events": round(launch_ms, 5),
            },
        }
        if n_gpus == 1 and not args.no_cpu_baseline and args.dtype == "f32":
            result["cpu_baseline"] = cpu_baseline(shape, bases)
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
