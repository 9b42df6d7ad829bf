#!/usr/bin/env python3
"""Does the second pass of an N-D transform find its input in the 256-MiB Infinity Cache?  Footprint sweep (GPU box):
batches of 128^3 volumes (16.8 MB each) and of 640 x 480 images (2.46 MB each), time per unit as the tensor grows past
the cache, with the cache policy on (MIFFT_ND_CACHE=3, default) and off (0).   python tools/ic_footprint_probe.py"""
import os
os.environ.setdefault("MIFFT_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hackathon_fft_amd", "csrc", "libmifft_lab.so"))  # the MIFFT_* switches below exist in the lab build only
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
import hackathon_fft_amd as mf
shape = %r
x = torch.randn(tuple(shape) + (2,), device="cuda")
out = torch.empty_like(x)
ctx = mf.DeviceContext(0)
plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
for _ in range(30): mf.fft(out, x, ctx, plan=plan)
torch.cuda.synchronize()
print(mf.time_fft(out, x, plan=plan, iters=100, ctx=ctx), [plan.kernel_name(d) for d in range(len(shape) - 1)])
"""


def run(shape, mode):
    e = dict(os.environ, MIFFT_ND_CACHE=str(mode))
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, list(shape))], env=e, capture_output=True, text=True)
    if r.returncode:
        return None, r.stderr[-200:]
    ms, names = r.stdout.strip().split(" ", 1)
    return float(ms), names


for unit, dims, counts in (("128^3 volume", (128, 128, 128), (2, 4, 6, 8, 10, 12, 14, 16, 20, 30)),
                           ("640x480 image", (640, 480), (20, 40, 60, 80, 100, 110, 120, 140, 200))):
    print(f"--- {unit}: out tensor MB | us per unit, policy on | policy off | kernels")
    for b in counts:
        mb = b * 8 * 1.0
        for d in dims:
            mb *= d
        mb /= 1e6
        on, names = run((b,) + dims, 3)
        off, _ = run((b,) + dims, 0)
        if on is None or off is None:
            print(b, "failed", names)
            continue
        print(f"{b:4d}  {mb:7.1f} MB   {on * 1e3 / b:8.2f}   {off * 1e3 / b:8.2f}   {names}", flush=True)
