"""16384-point rows: the four-step inside one LDS plane (default) against MIFFT_ROW2D=0 (one workgroup per row with the
global twiddle table, or two column-tile launches for big batches).   python tools/row2d_probe.py"""
import os
os.environ.setdefault("MIFFT_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hackathon_fft_amd", "csrc", "libmifft_lab.so"))  # the MIFFT_* switches below exist in the lab build only
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import hackathon_fft_amd as mf
ctx = mf.DeviceContext(0)
for n in (16384, 8192):
  for batch in (1, 7, 100, 1000, 7812, 20000):
      full = (batch, n, 2)
      x = torch.randn(full, device="cuda:0"); out = torch.full_like(x, float("nan"))
      plan = mf.plan_fft(torch.float32, torch.float32, full, full, ctx=ctx)
      mf.fft(out, x, ctx, plan=plan); ctx.synchronize()
      k = min(batch, 3)
      ref = np.fft.fft(x[:k].cpu().numpy().astype(np.float64).view(np.complex128)[..., 0], axis=1)
      got = out[:k].cpu().numpy().astype(np.float64).view(np.complex128)[..., 0]
      err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
      inv = mf.plan_fft(torch.float32, torch.float32, full, full, inverse=True, ctx=ctx)
      back = torch.empty_like(x); mf.fft(back, out, ctx, plan=inv); ctx.synchronize()
      rt = ((back - x).norm() / x.norm()).item()
      mf.time_fft(out, x, plan=plan, iters=20, ctx=ctx)
      ms = min(mf.time_fft(out, x, plan=plan, iters=50, ctx=ctx) for _ in range(3))
      print(f"N {n} batch {batch:6d}  {ms:8.4f} ms  {x.numel()*8/ms/1e9:6.3f} TB/s  {plan.num_launches} {plan.kernel_name(0)}  err {err:.1e} roundtrip {rt:.1e} nan {bool(torch.isnan(out).any())}", flush=True)
      del x, out, back, plan, inv
""" % ROOT
for off in (True, False):
    env = dict(os.environ)
    if off:
        env["MIFFT_ROW2D"] = "0"
    print("MIFFT_ROW2D =", "0" if off else "default", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
