"""Kernel time of random 2-D / 3-D shapes on ~128-MB tensors, slowest first (GPU box):  python tools/nd_sweep.py [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hackathon_fft_amd as mf

def smooth(n, lim=31):
    d = 2
    while d * d <= n:
        while n % d == 0:
            n //= d
        d += 1
    return n <= lim

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
pool = [n for n in range(8, 1500) if smooth(n)]
rows = []
for i in range(40):
    nd = random.choice([2, 2, 3])
    dims = tuple(random.choice(pool if nd == 2 else [p for p in pool if p <= 200]) for _ in range(nd))
    per = 8
    for d in dims:
        per *= d
    batch = max(1, int(128e6 / per))
    x = torch.randn((batch,) + dims + (2,), device="cuda:0")
    out = torch.empty_like(x)
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
        mf.time_fft(out, x, plan=plan, iters=3, ctx=ctx)
        ms = mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
    gbs = 2.0 * x.numel() * 4 / ms / 1e6
    rows.append((gbs / plan.num_launches, f"{(batch,) + dims}: {ms:.4f} ms  {gbs:6.0f} GB/s over {plan.num_launches} launches  "
                 f"{[plan.kernel_name(d) for d in range(nd)]}"))
    del x, out
for _, line in sorted(rows)[:15]:
    print(line)
