set -o pipefail
run() { python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-compare-vendor --no-live-pmc 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('OK', d['config']['workload'], d['scaling'], d['dtype'], d['ms_per_step'], d['roofline']['frac'] if 'roofline' in d else '-')" || { echo "FAIL $@"; tail -3 /tmp/err.txt; }; }
run --scaling strong
run --faithful --no-configs --no-strong-leg
run --dtype f64 --no-configs --no-strong-leg
for w in 3d_100x64x64x64 3d_1x256x256x256 1d_100x16384 2d_10x1920x1080 2d_1x3840x2160 2d_1x7680x4320 1d_64x1048576_fourstep 1d_290kx343_radix7 1d_330kx97_prime 2d_3200x100x100_plane 1d_100kx1024_real 2d_100x640x480_real 3d_10x128x128x128_real; do run --workload $w --no-configs --no-strong-leg; done
