#!/usr/bin/env python3
"""Workgroups per CU of the persistent grid, per table kernel: times bench workloads with MIFFT_GRID_PER_CU = 0 (formula),
1..8 on the lab build of the library (hackathon_fft_amd/csrc/libmifft_lab.so), interleaved on one box.
    python tools/grid_sweep.py [workload ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "hackathon_fft_amd", "csrc", "libmifft_lab.so")
wl = sys.argv[1:] or ["1d_100kx1024_radix2", "1d_500kx128", "1d_500kx93_radix31x3", "2d_100x640x480", "3d_10x128x128x128"]
for w in wl:
    row = []
    for g in (0, 1, 2, 3, 4, 5, 6, 8, 0):
        env = dict(os.environ, MIFFT_LIBRARY=lib, MIFFT_GRID_PER_CU=str(g))
        if g == 0:
            env.pop("MIFFT_GRID_PER_CU")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--no-cpu-baseline", "--no-live-pmc",
                            "--no-copy-ceiling", "--steps", "100", "--warmup", "10"], env=env, capture_output=True, text=True)
        try:
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            row.append((g, d["ms_per_step"]))
        except Exception:
            row.append((g, None))
    print("%-24s " % w + "  ".join("%d:%s" % (g, ("%.4f" % ms) if ms else "fail") for g, ms in row), flush=True)
