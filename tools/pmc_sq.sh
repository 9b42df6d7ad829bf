#!/bin/bash
# SQ-side counters for one bench workload (GPU box): instruction mix, busy / wait cycles, LDS conflicts.
#   tools/pmc_sq.sh <tag> <workload>   -> gpurun_out/sq_<tag>/pass*/...counter_collection.csv
set -o pipefail
tag=${1:-x}
wl=${2:-2d_100x640x480}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/sq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pass$i -- python3 $root/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-strong-leg --no-compare-vendor --no-live-pmc --no-copy-ceiling > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; exit 1; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mifft" not in k: continue
        acc[k[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
