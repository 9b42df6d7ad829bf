#!/bin/bash
export MIFFT_LIBRARY=${MIFFT_LIBRARY:-$PWD/hackathon_fft_amd/csrc/libmifft_lab.so}  # MIFFT_ND_CACHE is a lab-build switch
# round-2 measurement batch 1 (GPU box)
set -o pipefail
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2
mkdir -p $out
cd $root
for g in 17 18; do timeout -k 10 300 tools/tune/tune_tile_g$g > $out/tune_g$g.txt 2>&1 || echo "tune $g failed"; done
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q > $out/pytest_dist.log 2>&1; echo "pytest dist rc=$?"
timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
for wl in 2d_100x640x480 3d_10x128x128x128; do
  for m in 0 1 3; do
    export MIFFT_ND_CACHE=$m
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_${wl}_m$m -- python3 $root/bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline --no-compare-vendor > $out/trace_${wl}_m$m.log 2>&1 || echo "trace $wl $m failed"
    f=$(ls -t $out/trace_${wl}_m$m/*/*kernel_stats.csv 2>/dev/null | head -1)
    echo "== $wl mode $m"; [ -n "$f" ] && cut -d, -f1-7 "$f" | head -4 | cut -c1-200
  done
done
unset MIFFT_ND_CACHE
