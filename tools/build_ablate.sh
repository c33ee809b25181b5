#!/bin/bash
# Diagnostic builds of the library with parts of k_sweep_band removed (DESC_BAND_ABLATE bit mask, see pgd.hip):
#   tools/build_ablate.sh 1 2 8 ...   ->  tools/probes/libdesc_amd_abl<N>.so   (use with DESC_AMD_LIB=...)
cd "$(dirname "$0")/.."
for n in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function -pthread \
        -DDESC_BAND_ABLATE=$n -I include -I desc_amd/csrc -o tools/probes/libdesc_amd_abl$n.so desc_amd/csrc/*.cpp desc_amd/csrc/*.hip &
done
wait
