#!/bin/bash
for wl in C2 C3 C4 C5; do for t in 1 0; do echo "$wl TIGHT=$t"; DESC_DEBUG_SPECTRAL_TIGHT=$t timeout 300 python3 tools/spectral_laps.py $wl 2>&1 | grep -E "outer|spectral ms|gcw ms" | awk '/outer/{last=$0} /ms/{print last; print $0}' | tail -4; done; done
