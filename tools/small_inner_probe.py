import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, hackathon_fft_amd as mf
for shape in ((6400, 1024, 3), (3000, 640, 8), (2000, 128, 5, 2), (500, 343, 7), (40000, 64, 2), (999, 97, 3)):
    x = torch.randn(shape + (2,), device="cuda:0"); out = torch.full_like(x, float("nan"))
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
        mf.fft(out, x, ctx, plan=plan); ctx.synchronize()
        got = torch.view_as_complex(out[-3:].contiguous()).cpu().numpy()
        ref = np.fft.fftn(torch.view_as_complex(x[-3:].contiguous()).cpu().numpy().astype(np.complex128), axes=tuple(range(1, len(shape))))
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        ms = mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
        print(f"{shape}: {ms:.4f} ms  {[plan.kernel_name(d) for d in range(len(shape)-1)]}  err {err:.2e} nan {bool(torch.isnan(out).any())}")
