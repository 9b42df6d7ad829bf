"""Plan-creation latency (host twiddle tables, kernel selection, hipRTC where needed) per shape; second creation of the
same shape in the same process beside it.   python tools/plan_time_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hackathon_fft_amd as mf  # noqa: E402

ctx = mf.DeviceContext(0)
torch.zeros(1, device="cuda:0")
for shape in [(100000, 1024), (500000, 93), (100, 640, 480), (10, 128, 128, 128), (100, 16384), (64, 1 << 20), (1, 1 << 24),
              (1, 7680, 4320), (290000, 343), (8, 100, 100)]:
    full = tuple(shape) + (2,)
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        plan = mf.plan_fft(torch.float32, torch.float32, full, full, ctx=ctx)
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        names = [plan.kernel_name(d) for d in range(len(shape) - 1)]
        del plan
    print(f"{str(shape):>24}  first {ts[0]:9.2f} ms   again {ts[1]:8.2f} ms   {names}", flush=True)
