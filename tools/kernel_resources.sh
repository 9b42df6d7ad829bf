#!/bin/bash
# Prints VGPR / SGPR / scratch / LDS / occupancy for every kernel in a .hip file (gfx950).
f=${1:-hackathon_fft_amd/csrc/kernels_fast.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/_kr.o 2>&1 \
 | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed -e 's/.*remark: //' -e 's/ \[-Rpass.*//' \
 | awk '/Function Name/{name=$NF} / VGPRs:/{v=$NF} /SGPRs:/{s=$NF} /ScratchSize/{sc=$NF} /Occupancy/{o=$NF} /LDS Size/{printf "%-110s VGPR %3s SGPR %3s scratch %4s occ %s\n", substr(name,1,110), v, s, sc, o}'
