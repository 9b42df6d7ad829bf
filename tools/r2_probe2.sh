#!/bin/bash
# round-2 measurement batch 2 (GPU box): wide column tiles, strided four-step
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r2
mkdir -p $out
cd $root
timeout -k 10 900 python -m pytest tests/test_gpu_generated_table.py tests/test_gpu_jit.py tests/test_gpu_parity.py -x -q > $out/pytest_wide.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_wide.log
for w in 2d_1x7680x4320 2d_1x3840x2160 2d_10x1920x1080 3d_100x64x64x64 3d_1x256x256x256 2d_100x640x480 3d_10x128x128x128 1d_64x1048576_fourstep; do
  python bench.py --workload $w --steps 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernels'], d['config']['launches_per_step'])"
done
