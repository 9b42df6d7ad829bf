import time, torch, sys
sys.path.insert(0, '/root/repo')
import hackathon_fft_amd as mf
from hackathon_fft_amd import _lib
ctx = mf.DeviceContext(0)
# host overhead: tiny problem
x = torch.randn(4, 128, 2, device='cuda'); o = torch.empty_like(x)
p = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=ctx)
for _ in range(100): mf.fft(o, x, ctx, plan=p)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(2000): mf.fft(o, x, ctx, plan=p)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("tiny: host issue %.2f us/call, total %.2f us/call" % ((t1 - t) / 2000 * 1e6, (t2 - t) / 2000 * 1e6))
# raw ctypes call without python checks
L = _lib.lib(); s = ctx.stream.cuda_stream
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(2000): L.mifft_exec(p._h, x.data_ptr(), o.data_ptr(), s)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("tiny raw ctypes: host issue %.2f us/call, total %.2f us/call" % ((t1 - t) / 2000 * 1e6, (t2 - t) / 2000 * 1e6))
# big problem: python loop vs in-library loop
x = torch.randn(100000, 1024, 2, device='cuda'); o = torch.empty_like(x)
p = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, bases=[[2]], ctx=ctx)
for _ in range(50): mf.fft(o, x, ctx, plan=p)
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(200): mf.fft(o, x, ctx, plan=p)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ms_lib = mf.time_fft(o, x, plan=p, iters=200, ctx=ctx)
    print("big: python loop %.4f ms/step, in-library event loop %.4f ms/step" % ((t2 - t) / 200 * 1e3, ms_lib))
