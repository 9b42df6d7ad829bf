#!/bin/bash
# Runs on the GPU box (via gpurun).  Collects, for the headline workload:
#   1. rocprofv3 --kernel-trace --stats of `python bench.py` (per-kernel average duration)
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC slots), plus the same two
#      passes over tools/micro/copy_shapes, whose float2 row kernel moves a KNOWN byte count in the
#      same access pattern (calibration of the gfx950 FETCH_SIZE under-count).
# Raw output lands in gpurun_out/prof_<tag>_<workload>/; tools/summarize_prof.py condenses it into profiles/.
set -o pipefail
tag=${1:-r01}
wl=${2:-1d_100kx1024_radix2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_${tag}_$wl
rm -rf $out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
FLAGS="--workload $wl --steps 100 --warmup 20 --no-cpu-baseline --no-configs --no-strong-leg --no-compare-vendor --no-live-pmc --no-copy-ceiling"
# 0. the same command un-profiled on THIS box: the ms_per_step the profiled kernel durations must stay below
python3 $root/bench.py $FLAGS > $out/bench_unprofiled.json 2> $out/bench_unprofiled.err || { echo "bench failed"; tail -5 $out/bench_unprofiled.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py $FLAGS > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python3 $root/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-strong-leg --no-compare-vendor --no-live-pmc --no-copy-ceiling > $out/pmc_$c.log 2>&1 || { echo "pmc $c failed"; tail -5 $out/pmc_$c.log; exit 1; }
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/cal_$c -- $root/tools/micro/copy_shapes > $out/cal_$c.log 2>&1 || { echo "cal $c failed"; tail -5 $out/cal_$c.log; exit 1; }
done
ls $out
