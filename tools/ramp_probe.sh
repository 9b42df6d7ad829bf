#!/bin/bash
# How much does the K = 20 timed window of the driver's bench command jitter, and does a longer clock ramp help?
#   tools/ramp_probe.sh   (GPU box): six runs per ramp length, ms_per_step of the timed steps and of the HIP-event loop
for ramp in 0.05 0.2 0.5 0.05 0.2 0.5; do
  for i in 1 2 3; do
    MIFFT_BENCH_RAMP_S=$ramp python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-strong-leg --no-compare-vendor --no-live-pmc --no-copy-ceiling 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ramp $ramp  timed %.5f  events %.5f' % (d['ms_per_step'], d['roofline']['launch_ms_hip_events']))"
  done
done
