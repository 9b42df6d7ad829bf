#!/bin/bash
# Runs every bench workload once (GPU box) and prints one line per workload: ms per exec, roofline fraction, kernels.
for w in 1d_100kx1024_radix2 1d_500kx93_radix31x3 1d_500kx128 2d_100x640x480 3d_10x128x128x128 3d_100x64x64x64 \
         3d_1x256x256x256 1d_100x16384 2d_10x1920x1080 2d_1x3840x2160 2d_1x7680x4320 1d_64x1048576_fourstep \
         1d_290kx343_radix7 1d_330kx97_prime 2d_3200x100x100_plane; do
  timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-26s %8.4f ms  frac %.3f  %s' % (d['config']['workload'], d['ms_per_step'], d['roofline']['frac'], d['config']['kernels']))"
done
