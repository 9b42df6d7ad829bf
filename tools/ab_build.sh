#!/bin/bash
# Builds an A/B variant of libmifft.so into build_alt/<tag>/ (git-ignored, shipped to the GPU box by gpurun).
#   tools/ab_build.sh <tag> <extra hipcc flags, e.g. -DMIFFT_ALT_ROWS480=1> [-- <objects to rebuild, default kernels_fast>]
set -e
tag=$1; shift
flags=(); objs=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; objs=("$@"); break; fi
  flags+=("$1"); shift
done
[ ${#objs[@]} -eq 0 ] && objs=(kernels_fast)
src=hackathon_fft_amd/csrc
make -C $src -j8 >/dev/null
mkdir -p build_alt/$tag
for o in "${objs[@]}"; do
  ext=hip; [ -f $src/$o.cpp ] && ext=cpp
  x=(); [ $ext = cpp ] && x=(-x hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "${flags[@]}" "${x[@]}" -c $src/$o.$ext -o build_alt/$tag/$o.o
done
all=()
for f in mifft_api planner kernels_jit kernels_generic kernels_fast kernels_fourstep kernels_fast_gen_rows kernels_fast_gen_cols kernels_fast_gen_rows_f64 kernels_fast_gen_cols_f64; do
  if [ -f build_alt/$tag/$f.o ]; then all+=(build_alt/$tag/$f.o); else all+=($src/$f.o); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_alt/$tag/libmifft.so "${all[@]}" -L/opt/rocm/lib -lhiprtc
echo built build_alt/$tag/libmifft.so
