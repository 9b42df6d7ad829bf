import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, hackathon_fft_amd as mf
for n, dt in ((4984, torch.float64), (6000, torch.float64), (8192, torch.float64), (5120, torch.float64), (12000, torch.float32), (10000, torch.float32), (15625, torch.float32)):
    batch = max(1, int(64e6 / (n * (16 if dt == torch.float64 else 8))))
    x = torch.randn((batch, n, 2), device="cuda:0", dtype=dt); out = torch.full_like(x, float("nan"))
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(dt, dt, x.shape, x.shape, ctx=ctx)
        mf.fft(out, x, ctx, plan=plan); ctx.synchronize()
        got = torch.view_as_complex(out[-2:].contiguous()).cpu().numpy()
        ref = np.fft.fft(torch.view_as_complex(x[-2:].contiguous()).cpu().numpy().astype(np.complex128), axis=1)
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        ms = mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
        print(f"N {n} {dt}: {ms:.4f} ms per 64 MB  {plan.kernel_name(0)}  err {err:.2e}")
