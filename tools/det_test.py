import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
import hackathon_fft_amd as mf
rng = np.random.default_rng(11)
x = torch.from_numpy(rng.standard_normal((10, 1024, 2)).astype(np.float32)).cuda()
ctx = mf.DeviceContext(0)
plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
print(plan.kernel_name(0))
outs = []
for i in range(4):
    o = torch.full_like(x, float('nan')); mf.fft(o, x, ctx, plan=plan); ctx.synchronize(); outs.append(o.cpu().numpy())
for i in range(1, 4):
    print("rerun equal:", np.array_equal(outs[0], outs[i]), np.abs(outs[0]-outs[i]).max())
o = torch.full_like(x, float('nan')); mf.fft(o, x, ctx, plan=plan, first=3, count=4); ctx.synchronize(); p = o.cpu().numpy()
d = np.abs(p[3:7] - outs[0][3:7]); print("slab max diff", d.max(), "argmax row", np.unravel_index(d.argmax(), d.shape), "rows differing", [int((d[r] > 0).sum()) for r in range(4)])
# same row placed at different tile slots
xs = x[3:4].repeat(10, 1, 1).contiguous()
o = torch.full_like(xs, float('nan')); mf.fft(o, xs, ctx, plan=plan); ctx.synchronize(); q = o.cpu().numpy()
print("same row in 10 slots: diffs vs slot0:", [float(np.abs(q[i]-q[0]).max()) for i in range(10)])
