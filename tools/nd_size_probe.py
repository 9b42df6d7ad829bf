"""The N-D cache policy (MIFFT_ND_CACHE) at three tensor sizes per shape family: fully cache-resident,
around the cache size, beyond it.   python tools/nd_size_probe.py"""
import os
os.environ.setdefault("MIFFT_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hackathon_fft_amd", "csrc", "libmifft_lab.so"))  # the MIFFT_* switches below exist in the lab build only
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
import hackathon_fft_amd as mf
ctx = mf.DeviceContext(0)
for dims in [(640, 480), (128, 128, 128), (64, 64, 64)]:
    per = 8
    for d in dims: per *= d
    for mb in (60, 120, 150, 170, 200, 245):
        batch = max(1, int(mb * 1e6 / per))
        full = (batch,) + dims + (2,)
        x = torch.randn(full, device="cuda:0"); out = torch.empty_like(x)
        plan = mf.plan_fft(torch.float32, torch.float32, full, full, ctx=ctx)
        mf.time_fft(out, x, plan=plan, iters=20, ctx=ctx)
        ms = min(mf.time_fft(out, x, plan=plan, iters=50, ctx=ctx) for _ in range(3))
        print(f"{str(dims):>16} {batch*per/1e6:6.0f} MB per tensor  {ms:8.4f} ms  {ms/batch*1e3:8.3f} us/transform  {[plan.kernel_name(d) for d in range(len(dims))][-1]}", flush=True)
        del x, out, plan
""" % ROOT
for mode in ("0", "3"):
    env = dict(os.environ, MIFFT_ND_CACHE=mode)
    print("MIFFT_ND_CACHE =", mode, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
