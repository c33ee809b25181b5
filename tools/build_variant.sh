#!/bin/bash
# A/B builds of the library with extra macros:  tools/build_variant.sh <tag> -DX=1 ...  ->  tools/probes/libdesc_amd_<tag>.so
# compare on the GPU box with tools/lib_ab.sh tools/probes/libdesc_amd_<tag>.so - C2 C4 C5
cd "$(dirname "$0")/.."
tag=$1; shift
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function -pthread \
      "$@" -I include -I desc_amd/csrc -o tools/probes/libdesc_amd_$tag.so desc_amd/csrc/*.cpp desc_amd/csrc/*.hip
