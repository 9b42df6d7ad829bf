#!/usr/bin/env python3
"""Infinity-Cache policy of the N-D transforms (GPU box): MIFFT_ND_CACHE = 0 plain, 1 non-temporal loads of x in
the first pass, 2 in-place passes walk their tiles in alternating directions, 3 both.  One bench.py subprocess per
(workload, mode) -- the policy is fixed at plan creation -- plus a bit-exactness check of every mode against mode 0.

    python tools/nd_cache_probe.py [workload ...]
"""
import hashlib
import json
import os
os.environ.setdefault("MIFFT_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hackathon_fft_amd", "csrc", "libmifft_lab.so"))  # the MIFFT_* switches below exist in the lab build only
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = sys.argv[1:] or ["2d_100x640x480", "3d_10x128x128x128", "3d_100x64x64x64", "3d_1x256x256x256",
                             "2d_10x1920x1080"]
MODES = [0, 1, 2, 3]

CHECK = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
import hackathon_fft_amd as mf
shape = %r
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn(tuple(shape) + (2,), generator=g, device="cuda")
out = torch.full_like(x, float("nan"))
plan = mf.plan_fft(torch.float32, torch.float32, x.shape, x.shape, ctx=mf.DeviceContext(0))
mf.fft(out, x, plan=plan)
torch.cuda.synchronize()
print(hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest(), [plan.kernel_name(d) for d in range(len(shape) - 1)])
"""


def run(cmd, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, cwd=ROOT)
    if r.returncode:
        print("FAILED", cmd, r.stderr[-2000:])
        return None
    return r.stdout.strip().splitlines()[-1]


def main():
    sys.path.insert(0, ROOT)
    import bench
    for wl in WORKLOADS:
        shape = bench.WORKLOADS[wl][0]
        ref = None
        for m in MODES:
            env = {"MIFFT_ND_CACHE": str(m)}
            chk = run([sys.executable, "-c", CHECK % (ROOT, list(shape))], env)
            if m == 0:
                ref = chk.split()[0] if chk else None
            same = bool(chk) and chk.split()[0] == ref
            line = run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "300", "--warmup",
                        "20", "--no-cpu-baseline"], env)
            if not line:
                continue
            d = json.loads(line)
            print(f"{wl:24s} mode {m}: {d['ms_per_step']:.4f} ms/step  events {d['roofline']['launch_ms_hip_events']:.4f} ms  "
                  f"frac {d['roofline']['frac']:.3f}  bit-identical-to-mode-0 {same}  {d['config']['kernels']}", flush=True)


if __name__ == "__main__":
    main()
