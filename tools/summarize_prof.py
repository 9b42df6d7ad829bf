#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>_<workload>/ (rocprofv3 CSVs) into profiles/<tag>_<workload>.json + .md.

Kernel durations: from the kernel TRACE of `python3 bench.py --workload W --steps K --warmup W ...`, only the
launches of the K TIMED steps are averaged -- bench.py prints how many execs its clock ramp ran, so the launches of
ramp + warm-up (clocks and caches still settling; they raised the round-1 averages above the un-profiled
ms_per_step) and of the trailing mifft_time_exec loop are dropped by position.  The un-profiled ms_per_step of the same
command on the same box is stored beside them: sum of the timed-step kernel averages <= ms_per_step must hold.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE come from separate
passes, are in KiB, and FETCH_SIZE under-counts wide coalesced reads on gfx950 -- the correction factor is calibrated
on a kernel of the same access pattern that moves a known byte count (tools/micro/copy_shapes.hip, `copy_row_f2`:
819.2 MB read + 819.2 MB written per launch)."""
import csv
import glob
import json
import os
import sys
from collections import OrderedDict, defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
wl = sys.argv[2] if len(sys.argv) > 2 else "1d_100kx1024_radix2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}_{wl}")
HBM_PEAK = 8.0e12


def is_ours(name):
    return "mifft::" in name


def short(name):
    name = name.replace("void ", "")
    if is_ours(name) or "copy_" in name:
        return name[:160]
    return name[:60] + "..."


def newest(pattern):
    """gpurun merges every run into the same directory: keep only the most recent file of a pass"""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []


def bench_line(path):
    try:
        for ln in open(path):
            if ln.startswith("{"):
                return json.loads(ln)
    except Exception:
        pass
    return None


def timed_launches(d, line):
    """durations (us) of our kernels during the K timed steps, keyed by kernel name in launch order"""
    rows = []
    for f in newest(os.path.join(d, "**", "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            if is_ours(r["Kernel_Name"]):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    L = line["config"]["launches_per_step"]
    first = (line["ramp"]["execs"] + line["warmup"]) * L
    last = first + line["steps"] * L
    if len(rows) < last:
        return None, len(rows)
    per = OrderedDict()
    for s, e, n in rows[first:last]:
        per.setdefault(n, []).append((e - s) / 1e3)
    return per, len(rows)


def pmc(d, counter):
    """average counter value per dispatch, per kernel"""
    acc = defaultdict(list)
    for f in newest(os.path.join(d, "**", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


line = bench_line(os.path.join(src, "trace.log"))
unprof = bench_line(os.path.join(src, "bench_unprofiled.json"))
if line is None:
    sys.exit(f"no bench JSON line in {src}/trace.log")
per, n_all = timed_launches(os.path.join(src, "trace"), line)
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
if per:
    for name, us in per.items():
        e = {"kernel": name, "timed_launches": len(us), "launches_per_step": len(us) // line["steps"],
             "avg_us": sum(us) / len(us), "min_us": min(us), "max_us": max(us)}
        f, w = fetch.get(name), write.get(name)
        if f is not None and w is not None and cal.get("fetch_factor"):
            e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = f, w
            e["hbm_read_bytes_corrected"] = f * 1024.0 * cal["fetch_factor"]
            e["hbm_write_bytes_corrected"] = w * 1024.0 * cal["write_factor"]
            e["hbm_traffic_bytes"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes_corrected"]
        kernels.append(e)

algo = line["roofline"]["algorithmic_bytes_per_launch"]
step_us = sum(k["avg_us"] * k["launches_per_step"] for k in kernels) if kernels else None
res = {
    "tag": tag, "workload": wl,
    "command": f"python3 bench.py --workload {wl} --steps {line['steps']} --warmup {line['warmup']} --no-cpu-baseline "
               "--no-configs --no-strong-leg --no-compare-vendor",
    "bench_kernels": sorted(set(line["config"]["kernels"])),
    "launches_in_trace": n_all, "ramp_execs": line["ramp"]["execs"], "warmup": line["warmup"], "steps": line["steps"],
    "timed_step_kernel_us": step_us,
    "roofline_frac_from_profile": (algo / (step_us * 1e-6) / HBM_PEAK) if step_us else None,
    "profiled_run": {"ms_per_step": line["ms_per_step"], "launch_ms_hip_events": line["roofline"]["launch_ms_hip_events"],
                     "frac": line["roofline"]["frac"]},
    "unprofiled_run_same_box": ({"ms_per_step": unprof["ms_per_step"],
                                 "launch_ms_hip_events": unprof["roofline"]["launch_ms_hip_events"],
                                 "frac": unprof["roofline"]["frac"]} if unprof else None),
    "calibration": cal, "kernels": kernels,
}
if step_us and unprof:
    res["kernel_time_within_unprofiled_step"] = bool(step_us / 1e3 <= unprof["ms_per_step"] * 1.02)
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
base = os.path.join(root, "profiles", f"{tag}_{wl}")
json.dump(res, open(base + ".json", "w"), indent=1)
with open(base + ".md", "w") as f:
    f.write(f"# rocprofv3 summary {tag} / {wl}\n\n")
    f.write(f"command: `rocprofv3 --kernel-trace --stats -- {res['command']}`\n\n")
    f.write(f"{n_all} launches of our kernels in the trace; the {line['steps']} timed steps are launches "
            f"{(line['ramp']['execs'] + line['warmup']) * line['config']['launches_per_step']}.. "
            f"(after {line['ramp']['execs']} ramp execs + {line['warmup']} warm-up steps).\n\n")
    f.write("| kernel (timed steps only) | launches | avg us | min us | max us |\n|---|---|---|---|---|\n")
    for k in kernels:
        f.write(f"| `{k['kernel'][:120]}` | {k['timed_launches']} | {k['avg_us']:.2f} | {k['min_us']:.2f} | {k['max_us']:.2f} |\n")
    if step_us:
        f.write(f"\nkernel time per step {step_us:.2f} us -> {algo / 1e6:.1f} MB / {step_us:.2f} us / 8 TB/s = "
                f"**{res['roofline_frac_from_profile']:.4f}**; profiled run's own line: ms_per_step {line['ms_per_step']}, "
                f"HIP-event launch ms {line['roofline']['launch_ms_hip_events']} (frac {line['roofline']['frac']})")
        if unprof:
            f.write(f"; un-profiled run on the same box: ms_per_step {unprof['ms_per_step']}, HIP-event launch ms "
                    f"{unprof['roofline']['launch_ms_hip_events']} (frac {unprof['roofline']['frac']})")
        f.write(".\n")
    f.write("\nPMC (separate passes, per launch):\n\n")
    f.write(f"calibration on copy_row_f2 (known 819.2 MB each way): {json.dumps(cal)}\n\n")
    for k in kernels:
        if "hbm_traffic_bytes" in k:
            f.write(f"* `{k['kernel'][:100]}`: FETCH_SIZE {k['FETCH_SIZE_KiB']:.0f} KiB, WRITE_SIZE {k['WRITE_SIZE_KiB']:.0f} KiB -> "
                    f"corrected HBM traffic {k['hbm_traffic_bytes'] / 1e6:.1f} MB per launch\n")
print(json.dumps(res, indent=1)[:3000])
