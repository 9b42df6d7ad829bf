"""Every shape of the reference's own benchmark list (fft/bench.mojo:108-124, the active and the commented-out ones),
complex input and REAL input (the reference benchmarks `bench_gpu_radix_n_rfft`: real in, full complex spectrum out),
kernel time by HIP events, fraction of the 8 TB/s roofline on the algorithmic bytes (8 B or 4 B in + 8 B out per point),
rocFFT C2C beside it where rocFFT has the rank (<= 3 dims).   python tools/ref_shapes_sweep.py [--no-vendor]"""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hackathon_fft_amd as mf  # noqa: E402

SHAPES = [(250_000, 93), (250_000, 128), (100_000, 1024), (100, 16384), (100, 640, 480), (10, 1920, 1080),
          (1, 3840, 2160), (1, 7680, 4320), (100, 64, 64, 64), (10, 128, 128, 128), (1, 256, 256, 256),
          (1, 512, 512, 512), (1, 64, 64, 64, 64), (1, 25, 160, 160, 48)]


def main():
    vendor = {}
    exe = os.path.join(ROOT, "tools", "vendor_fft_bench")
    if "--no-vendor" not in sys.argv and os.path.exists(exe):
        specs = ["x".join(str(v) for v in s) for s in SHAPES if len(s) <= 4]
        r = subprocess.run([exe, "--iters", "50"] + specs, capture_output=True, text=True, timeout=600)
        for ln in r.stdout.splitlines():
            if ln.startswith("{"):
                d = json.loads(ln)
                vendor[d["shape"]] = d["ms"]
    print(f"{'shape':>22} {'in':>4} {'ms':>8} {'frac':>6} {'rocFFT ms':>10}  kernels")
    for shape in SHAPES:
        for comps in (2, 1):
            x = torch.randn(shape + (comps,), device="cuda:0")
            out = torch.empty(shape + (2,), device="cuda:0")
            with mf.DeviceContext(0) as ctx:
                plan = mf.plan_fft(torch.float32, torch.float32, x.shape, out.shape, ctx=ctx)
                mf.time_fft(out, x, plan=plan, iters=10, ctx=ctx)
                ms = min(mf.time_fft(out, x, plan=plan, iters=30, ctx=ctx) for _ in range(3))
            pts = out.numel() // 2
            frac = pts * (4 * comps + 8) / (ms * 1e-3) / 8e12
            spec = "x".join(str(v) for v in shape)
            v = vendor.get(spec) if comps == 2 else None
            print(f"{spec:>22} {'c2c' if comps == 2 else 'r2c':>4} {ms:8.4f} {frac:6.3f} {v if v is not None else '':>10}  "
                  f"{plan.num_launches} launches {[plan.kernel_name(d) for d in range(len(shape) - 1)]}", flush=True)
            del x, out, plan


if __name__ == "__main__":
    main()
