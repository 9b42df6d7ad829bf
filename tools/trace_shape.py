"""Run one shape a few times so that `rocprofv3 --kernel-trace --stats -- python3 tools/trace_shape.py 1x4320x7680` shows
which kernels a plan launches and how long each takes.  usage: trace_shape.py BxD0[xD1[xD2]] [execs] [real]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hackathon_fft_amd as mf  # noqa: E402

shape = tuple(int(v) for v in sys.argv[1].split("x"))
execs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
full = shape + (2,)
real = len(sys.argv) > 3 and sys.argv[3] == "real"
x = torch.randn(shape + ((1,) if real else (2,)), device="cuda:0")
out = torch.empty(full, device="cuda:0")
ctx = mf.DeviceContext(0)
plan = mf.plan_fft(x.dtype, out.dtype, tuple(x.shape), full, ctx=ctx)
print("kernels", [plan.kernel_name(d) for d in range(len(shape) - 1)], "launches", plan.num_launches)
for _ in range(execs):
    mf.fft(out, x, ctx, plan=plan)
ctx.synchronize()
