"""fp64 side of tools/nts_probe.py: the non-temporal-store window on (default) and off (MIFFT_NTS_MIN_BYTES=1e18,
MIFFT_JIT_NT=0).   python tools/nts_probe_f64.py"""
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
for n in [128, 240, 343, 1000, 2000]:
    for total in [float(v) for v in __import__("os").environ.get("PROBE_TOTALS", "0.15e9,0.4e9,0.6e9").split(",")]:
        batch = int(total / 32 / n)
        full = (batch, n, 2)
        x = torch.randn(full, device="cuda:0", dtype=torch.float64); out = torch.empty_like(x)
        plan = mf.plan_fft(torch.float64, torch.float64, full, full, ctx=ctx)
        mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
        ms = min(mf.time_fft(out, x, plan=plan, iters=30, ctx=ctx) for _ in range(3))
        print(f"N {n:5d} total {total/1e9:4.2f} GB  {ms:8.4f} ms  {x.numel()*16/ms/1e9:6.3f} TB/s  {plan.kernel_name(0)}", flush=True)
        del x, out, plan
""" % ROOT
for off in (True, False):
    env = dict(os.environ)
    if off:
        env["MIFFT_NTS_MIN_BYTES"] = "1e18"
        env["MIFFT_JIT_NT"] = "0"
    print("window", "off" if off else "on", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
