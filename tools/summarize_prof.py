#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>_<workload>/ (rocprofv3 CSVs) into profiles/<tag>_<workload>.json + .md.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE come from
separate passes, are in KiB, and FETCH_SIZE under-counts wide coalesced reads on gfx950 -- the
correction factor is calibrated on a kernel of the same access pattern that moves a known byte count
(tools/micro/copy_shapes.hip, `copy_row_f2`: 819.2 MB read + 819.2 MB written per launch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
wl = sys.argv[2] if len(sys.argv) > 2 else "1d_100kx1024_radix2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}_{wl}")


def short(name):
    name = name.replace("void ", "")
    if "tile_kernel" in name or "generic_kernel" in name or "copy_" in name:
        return name[:160]
    return name[:60] + "..."


def newest(pattern):
    """gpurun merges every run into the same directory: keep only the most recent file of a pass"""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


def kernel_stats(d):
    rows = []
    for f in newest(os.path.join(d, "**", "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            rows.append({"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                         "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                         "pct": float(r["Percentage"])})
    return sorted(rows, key=lambda r: -r["pct"])


def pmc(d, counter):
    """average counter value per dispatch, per kernel"""
    acc = defaultdict(list)
    for f in newest(os.path.join(d, "**", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


stats = kernel_stats(os.path.join(src, "trace"))
fetch = pmc(os.path.join(src, "pmc_FETCH_SIZE"), "FETCH_SIZE")
write = pmc(os.path.join(src, "pmc_WRITE_SIZE"), "WRITE_SIZE")
cal_fetch = pmc(os.path.join(src, "cal_FETCH_SIZE"), "FETCH_SIZE")
cal_write = pmc(os.path.join(src, "cal_WRITE_SIZE"), "WRITE_SIZE")

known = 819.2e6
cal = {}
for k, v in cal_fetch.items():
    if "copy_row_f2" in k:
        cal["fetch_kib_reported"] = v
        cal["fetch_factor"] = known / (v * 1024.0)
for k, v in cal_write.items():
    if "copy_row_f2" in k:
        cal["write_kib_reported"] = v
        cal["write_factor"] = known / (v * 1024.0)

kernels = []
for s in stats:
    if not ("tile_kernel" in s["kernel"] or "generic_kernel" in s["kernel"]):
        continue
    e = dict(s)
    f = fetch.get(s["kernel"])
    w = write.get(s["kernel"])
    if f is not None and w is not None and cal.get("fetch_factor"):
        e["FETCH_SIZE_KiB"] = f
        e["WRITE_SIZE_KiB"] = w
        e["hbm_read_bytes_corrected"] = f * 1024.0 * cal["fetch_factor"]
        e["hbm_write_bytes_corrected"] = w * 1024.0 * cal["write_factor"]
        e["hbm_traffic_bytes"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes_corrected"]
    kernels.append(e)

res = {"tag": tag, "workload": wl, "calibration": cal, "kernels": kernels,
       "other_kernels": [s for s in stats if not ("tile_kernel" in s["kernel"] or "generic_kernel" in s["kernel"])][:5]}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
base = os.path.join(root, "profiles", f"{tag}_{wl}")
json.dump(res, open(base + ".json", "w"), indent=1)
with open(base + ".md", "w") as f:
    f.write(f"# rocprofv3 summary {tag} / {wl}\n\n")
    f.write("command: `rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --steps 100 --warmup 20 --no-cpu-baseline`\n\n" % wl)
    f.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
    for s in stats[:6]:
        f.write(f"| `{s['kernel'][:120]}` | {s['calls']} | {s['avg_us']:.2f} | {s['min_us']:.2f} | {s['max_us']:.2f} | {s['pct']:.2f} |\n")
    f.write("\nPMC (separate passes, per launch):\n\n")
    f.write(f"calibration on copy_row_f2 (known 819.2 MB each way): {json.dumps(cal)}\n\n")
    for k in kernels:
        if "hbm_traffic_bytes" in k:
            f.write(f"* `{k['kernel'][:100]}`: FETCH_SIZE {k['FETCH_SIZE_KiB']:.0f} KiB, WRITE_SIZE {k['WRITE_SIZE_KiB']:.0f} KiB -> "
                    f"corrected HBM traffic {k['hbm_traffic_bytes'] / 1e6:.1f} MB per launch\n")
print(json.dumps(res, indent=1)[:3000])
