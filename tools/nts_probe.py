"""Non-temporal STORES for batched 1-D rows of lengths outside the hand-tuned table, by tensor size:
MIFFT_JIT_NT=0 (plain kernels), default (runtime-specialised lengths store non-temporally above ~0.1 GB), MIFFT_JIT_NT=2
(lengths of the generated table too, through the runtime-specialised twin).   python tools/nts_probe.py"""
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
for n in [49, 96, 100, 121, 240, 343, 500, 1000, 1080, 2000, 3125]:
    for total in (0.15e9, 0.4e9, 0.9e9):
        batch = int(total / 16 / n)
        full = (batch, n, 2)
        x = torch.randn(full, device="cuda:0"); out = torch.empty_like(x)
        plan = mf.plan_fft(torch.float32, torch.float32, full, full, ctx=ctx)
        mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
        ms = min(mf.time_fft(out, x, plan=plan, iters=30, ctx=ctx) for _ in range(3))
        print(f"N {n:5d} total {total/1e9:4.2f} GB  {ms:8.4f} ms  {x.numel()*8/ms/1e9:6.3f} TB/s  {plan.kernel_name(0)}", flush=True)
        del x, out, plan
""" % ROOT
for knob in ("0", None, "2"):
    env = dict(os.environ)
    env.pop("MIFFT_JIT_NT", None)
    if knob:
        env["MIFFT_JIT_NT"] = knob
    print("MIFFT_JIT_NT =", knob, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
