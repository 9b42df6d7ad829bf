import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, hackathon_fft_amd as mf
for shape, comps, dt in [((10, 1080, 1920), 1, torch.uint8), ((100, 640, 480), 1, torch.uint8), ((100000, 1024), 2, torch.int32)]:
    x = torch.randint(0, 200, shape + (comps,), device="cuda:0").to(dt)
    out = torch.empty(shape + (2,), device="cuda:0")
    with mf.DeviceContext(0) as ctx:
        plan = mf.plan_fft(dt, torch.float32, x.shape, out.shape, ctx=ctx)
        mf.time_fft(out, x, plan=plan, iters=3, ctx=ctx)
        ms = mf.time_fft(out, x, plan=plan, iters=20, ctx=ctx)
        print(shape, comps, dt, "%.4f ms" % ms, [plan.kernel_name(d) for d in range(len(shape) - 1)])
