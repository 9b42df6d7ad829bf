#!/bin/bash
# the five BASELINE workloads (+ a few reference shapes) on each library build, interleaved: tools/ab_all.sh <tag> [<tag> ...]
for w in 1d_100kx1024_radix2 1d_500kx128 1d_500kx93_radix31x3 2d_100x640x480 3d_10x128x128x128 3d_100x64x64x64 3d_1x256x256x256 2d_10x1920x1080 1d_290kx343_radix7 ${AB_EXTRA}; do
  for tag in "$@"; do
    lib=build_alt/$tag/libmifft.so
    [ "$tag" = default ] && lib=hackathon_fft_amd/csrc/libmifft.so
    MIFFT_LIBRARY=$PWD/$lib timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline --no-live-pmc --no-copy-ceiling --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-14s %-24s %8.4f ms  ev %8.4f  %s' % ('$tag', d['config']['workload'], d['ms_per_step'], d['roofline']['launch_ms_hip_events'], d['config']['kernels']))"
  done
done
