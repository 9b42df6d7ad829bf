"""2-D plans whose only pass is a fused plane: non-temporal stores in the 0.25-0.65 GB window (default) against
plain stores (MIFFT_NTS_MIN_BYTES=1e18 MIFFT_JIT_NT=0).   python tools/plane_nts_probe.py"""
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
for dims in [(64, 64), (128, 128), (100, 100), (32, 32), (48, 80)]:
    for total in (0.15e9, 0.42e9, 0.9e9):
        batch = int(total / 16 / (dims[0] * dims[1]))
        full = (batch,) + dims + (2,)
        x = torch.randn(full, device="cuda:0"); out = torch.empty_like(x)
        plan = mf.plan_fft(torch.float32, torch.float32, full, full, ctx=ctx)
        mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
        ms = min(mf.time_fft(out, x, plan=plan, iters=30, ctx=ctx) for _ in range(3))
        print(f"{str(dims):>10} total {total/1e9:4.2f} GB  {ms:8.4f} ms  {x.numel()*8/ms/1e9:6.3f} TB/s  {plan.num_launches} {plan.kernel_name(1)}", flush=True)
        del x, out, plan
""" % ROOT
for off in (True, False):
    env = dict(os.environ)
    if off:
        env["MIFFT_NTS_MIN_BYTES"] = "1e18"
        env["MIFFT_JIT_NT"] = "0"
    print("window", "off" if off else "on", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
