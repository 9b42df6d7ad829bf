"""Kernel time of batched 1-D C2C transforms over a list of lengths (GPU box):
    python tools/size_sweep.py [--dtype f32|f64] [--cols INNER] N [N ...]
Each length runs on a ~256-MB tensor; prints ms per exec (HIP events inside the library), GB/s of algorithmic
traffic and the kernel the plan picked.  Use MIFFT_LIBRARY=<other build> for A/B runs of table changes."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hackathon_fft_amd as mf

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32")
ap.add_argument("--cols", type=int, default=0, help="transform dimension 0 of (N, INNER) images instead of rows")
ap.add_argument("--mb", type=float, default=256.0)
ap.add_argument("sizes", type=int, nargs="+")
a = ap.parse_args()
dt = torch.float32 if a.dtype == "f32" else torch.float64
esz = 8 if a.dtype == "f32" else 16
for n in a.sizes:
    if a.cols:
        batch = max(1, int(a.mb * 1e6 / (n * a.cols * esz)))
        shape = (batch, n, a.cols, 2)
        bases = None
    else:
        batch = max(1, int(a.mb * 1e6 / (n * esz)))
        shape = (batch, n, 2)
    x = torch.randn(shape, device="cuda:0", dtype=dt)
    out = torch.empty_like(x)
    with mf.DeviceContext(0) as ctx:
        try:
            plan = mf.plan_fft(dt, dt, x.shape, x.shape, ctx=ctx)
        except mf.MifftError as e:
            print(f"N {n:6d}: {e}")
            continue
        mf.time_fft(out, x, plan=plan, iters=5, ctx=ctx)
        ms = mf.time_fft(out, x, plan=plan, iters=20, ctx=ctx)
        names = [plan.kernel_name(d) for d in range(len(shape) - 2)]
        gbs = 2.0 * x.numel() * x.element_size() / ms / 1e6
        print(f"N {n:6d} batch {batch:8d}: {ms:8.4f} ms  {gbs:7.0f} GB/s  launches {plan.num_launches}  {names}")
    del x, out
