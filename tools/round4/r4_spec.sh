#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_spectral.py tests/test_gpu_refine.py tests/test_gpu_fullsize_next_rows.py -x -q -m gpu > gpurun_out/r4_tests_spec.log 2>&1; echo "spectral tests rc=$?"; tail -5 gpurun_out/r4_tests_spec.log
timeout -k 10 300 python3 tools/spectral_laps.py C4 > gpurun_out/r04_spectral_laps_c4.txt 2>&1; grep -E "outer|subspace|ms|products" gpurun_out/r04_spectral_laps_c4.txt | tail -40
timeout -k 10 300 python3 tools/next_rows_bench.py --workload C4 > gpurun_out/r04_c4_next_rows.json 2> gpurun_out/r04_c4_next_rows.err; tail -c 1800 gpurun_out/r04_c4_next_rows.json
