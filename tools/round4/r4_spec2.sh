#!/bin/bash
for f in 3 4 6 8; do echo "FIRST=$f"; DESC_DEBUG_SPECTRAL_FIRST=$f timeout 300 python3 tools/spectral_laps.py C4 2>&1 | grep -E "outer 4|outer 5|outer 6|spectral ms|gcw ms" | tail -4; DESC_DEBUG_SPECTRAL_FIRST=$f timeout 300 python3 tools/spectral_laps.py C3 2>&1 | grep -E "outer [4-9]|spectral ms|gcw ms" | tail -4; done
DESC_DEBUG_SPECTRAL_FIRST=4 timeout -k 10 600 python3 -m pytest tests/test_gpu_spectral.py tests/test_gpu_refine.py tests/test_gpu_fullsize_next_rows.py -x -q -m gpu 2>&1 | tail -3
