"""Strided four-step against the single column tile for strided dimensions of 1024..4096 points
(MIFFT_FS_STRIDED_MIN_N, tuning knob).   python tools/fs_minn_probe.py"""
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
for shape in [(1, 3840, 2160), (10, 1920, 1080), (4, 2048, 2048), (8, 4096, 512), (16, 1536, 1024), (2, 3072, 3072), (40, 1024, 1024), (6, 2560, 1440)]:
    full = shape + (2,)
    x = torch.randn(full, device="cuda:0"); out = torch.empty_like(x)
    plan = mf.plan_fft(torch.float32, torch.float32, full, full, ctx=ctx)
    mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
    ms = min(mf.time_fft(out, x, plan=plan, iters=30, ctx=ctx) for _ in range(3))
    print(f"{str(shape):>18} {ms:8.4f} ms  {plan.num_launches} launches {[plan.kernel_name(d) for d in range(2)]}", flush=True)
""" % ROOT
for knob in (None, "1024"):
    env = dict(os.environ)
    if knob:
        env["MIFFT_FS_STRIDED_MIN_N"] = knob
    print("MIFFT_FS_STRIDED_MIN_N =", knob, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
