#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -q -m gpu --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_fullsize_next_rows.py > gpurun_out/r4_tests_all4.log 2>&1; rc=$?; echo "tests (without the full-size files) rc=$rc"; tail -12 gpurun_out/r4_tests_all4.log
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_2.json 2> gpurun_out/r4_bench_2.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4_bench_2.json')); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['end_to_end']['ms'], d['end_to_end']['repeat_ms'], d['parity_vs_cpu']['max_abs'], d['secondary_config']['value'], d['secondary_config']['roofline']['frac'])"
