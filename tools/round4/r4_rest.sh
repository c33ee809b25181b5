#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -q -m gpu --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_fullsize_next_rows.py > gpurun_out/r4_tests_all3.log 2>&1; rc=$?; echo "tests (without the full-size files) rc=$rc"; tail -12 gpurun_out/r4_tests_all3.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fullsize_next_rows.py -q -m gpu > gpurun_out/r4_tests_fullsize3.log 2>&1; rc=$?; echo "full-size rc=$rc"; tail -12 gpurun_out/r4_tests_fullsize3.log
