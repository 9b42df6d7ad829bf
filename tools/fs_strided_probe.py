#!/usr/bin/env python3
"""Four-step of a long strided dimension (GPU box): the factorisation N = N1 * N2 forced through MIFFT_FS_N1, against the
planner's own (most balanced) choice and the transposed route.   python tools/fs_strided_probe.py [workload]"""
import json
import os
os.environ.setdefault("MIFFT_LIBRARY", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hackathon_fft_amd", "csrc", "libmifft_lab.so"))  # the MIFFT_* switches below exist in the lab build only
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl = sys.argv[1] if len(sys.argv) > 1 else "2d_1x7680x4320"


def run(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "100", "--no-cpu-baseline"],
                       env=e, capture_output=True, text=True)
    if r.returncode:
        return None
    d = json.loads(r.stdout.strip().splitlines()[-1])
    return d["ms_per_step"], d["config"]["kernels"], d["config"]["launches_per_step"]


print("planner's choice:", run({}))
print("transposed route:", run({"MIFFT_FOURSTEP_STRIDED": "0"}))
for n1 in (30, 40, 48, 60, 64, 80, 96, 120, 128, 160, 192, 240, 256):
    print("N1 =", n1, run({"MIFFT_FS_N1": str(n1)}), flush=True)
