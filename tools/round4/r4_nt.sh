#!/bin/bash
# round 4: cache-policy bits (nt) on the band sweep's stream loads / weight stores, A/B against the production build
for v in ldnt stnt bothnt; do
  bash tools/lib_ab.sh tools/probes/libdesc_amd_$v.so - C4
done
