#!/bin/bash
# Experimental builds of the library (DESC_EXP bit mask, see pgd.hip):  tools/build_exp.sh 1 2 ...  ->  tools/probes/libdesc_amd_exp<N>.so
# A/B against the production library on the GPU box with tools/lib_ab.sh tools/probes/libdesc_amd_exp<N>.so - C2 C4 C5
cd "$(dirname "$0")/.."
for n in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function -pthread \
        -DDESC_EXP=$n -I include -I desc_amd/csrc -o tools/probes/libdesc_amd_exp$n.so desc_amd/csrc/*.cpp desc_amd/csrc/*.hip &
done
wait
