import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, hackathon_fft_amd as mf
for n in (131, 251, 509, 1009, 2 * 251, 3 * 337, 2039, 4093, 2 * 2039, 5 * 1021):
    batch = max(1, int(64e6 / (n * 8)))
    x = torch.randn((batch, n, 2), device="cuda:0"); out = torch.empty_like(x)
    f, m, d = [], n, 2
    while m > 1:
        if m % d == 0:
            f.append(d)
            while m % d == 0: m //= d
        d += 1
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, bases=[f], ctx=ctx)
        mf.fft(out, x, ctx, plan=plan); ctx.synchronize()
        got = torch.view_as_complex(out[:4].contiguous()).cpu().numpy()
        ref = np.fft.fft(torch.view_as_complex(x[:4].contiguous()).cpu().numpy().astype(np.complex128), axis=1)
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        ms = mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
        print(f"N {n}: {ms:.4f} ms per 64 MB  {plan.kernel_name(0)}  err {err:.2e}")
