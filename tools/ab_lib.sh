#!/bin/bash
# A/B of kernel variants in ONE gpurun call (boxes differ by several %): builds of libmifft.so with different -D switches
# live in build_alt/<tag>/libmifft.so (tools/ab_build.sh); this script times one bench workload on each of them.
#   tools/ab_lib.sh <workload> <tag> [<tag> ...]        (tag "default" = the in-tree library)
w=$1; shift
for tag in "$@"; do
  lib=build_alt/$tag/libmifft.so
  [ "$tag" = default ] && lib=hackathon_fft_amd/csrc/libmifft.so
  for rep in 1 2; do
    MIFFT_LIBRARY=$PWD/$lib timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline --no-live-pmc --no-copy-ceiling --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-14s %-22s %8.4f ms  ev %8.4f  %s' % ('$tag', d['config']['workload'], d['ms_per_step'], d['roofline']['launch_ms_hip_events'], d['config']['kernels']))"
  done
done
