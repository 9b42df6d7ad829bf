"""Fused n x n planes (runtime-specialised plane_kernel) against separate row and column passes (MIFFT_JIT=0 keeps the
table kernels for table lengths).  GPU box:  python tools/plane_probe.py ; MIFFT_JIT=0 python tools/plane_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hackathon_fft_amd as mf

for n1, n in ((16, 16), (32, 32), (48, 48), (64, 64), (96, 96), (100, 100), (120, 120), (128, 128), (64, 128), (128, 64),
              (96, 160), (32, 512), (480, 32), (24, 16)):
    batch = max(1, int(256e6 / (n * n1 * 8)))
    x = torch.randn((batch, n1, n, 2), device="cuda:0")
    out = torch.empty_like(x)
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
        mf.time_fft(out, x, plan=plan, iters=3, ctx=ctx)
        ms = mf.time_fft(out, x, plan=plan, iters=20, ctx=ctx)
        print(f"{n1:4d} x {n:<4d} batch {batch:7d}: {ms:.4f} ms  launches {plan.num_launches}  {plan.kernel_name(1)}")
