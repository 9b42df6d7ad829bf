"""What does the ragged last round of a persistent grid cost?  Times row batches that are an exact number of rounds
(grid x tile rows) against batches just above and below.   python tools/tail_probe.py [N=93] [tile=64] [wgs_per_cu=3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hackathon_fft_amd as mf  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 93
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wpc = int(sys.argv[3]) if len(sys.argv) > 3 else 3
grid = 256 * wpc
rnd = grid * tile
target = int(sys.argv[4]) if len(sys.argv) > 4 else 500000
k = target // rnd
ctx = mf.DeviceContext(0)
for batch in [k * rnd, k * rnd + rnd // 8, k * rnd + rnd // 4, target, k * rnd + rnd // 2, k * rnd + 3 * rnd // 4, (k + 1) * rnd]:
    x = torch.randn(batch, n, 2, device="cuda:0")
    out = torch.empty_like(x)
    plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
    mf.time_fft(out, x, plan=plan, iters=20, ctx=ctx)
    ms = min(mf.time_fft(out, x, plan=plan, iters=50, ctx=ctx) for _ in range(4))
    print(f"batch {batch:8d} = {batch / rnd:6.3f} rounds  {ms:.4f} ms  {ms / batch * 1e6:.4f} ns/row  "
          f"{batch * n * 16 / ms / 1e9:.3f} TB/s  {plan.kernel_name(0)}", flush=True)
    del x, out, plan
