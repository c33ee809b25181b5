#!/bin/bash
# round 4: DESC() with the device problem uploaded under the structure build (helper thread in the Python wrapper): tests, then laps
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_refine.py tests/test_gpu_fullsize_next_rows.py -x -q -m gpu -k "DESC or desc or demo or refine or pipeline" 2>&1 | tail -2
timeout -k 10 300 python3 tools/wrapper_laps.py --workload C4 2>&1 | tail -8
