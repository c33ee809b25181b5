#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -q -m gpu --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_fullsize_next_rows.py -x > gpurun_out/r4_tests_all5.log 2>&1; rc=$?; echo "tests (without the full-size files) rc=$rc"; tail -8 gpurun_out/r4_tests_all5.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fullsize_next_rows.py -q -m gpu -x > gpurun_out/r4_tests_fullsize5.log 2>&1; rc=$?; echo "full-size rc=$rc"; tail -6 gpurun_out/r4_tests_fullsize5.log
[ $rc -ne 0 ] && exit 1
LAPS_LEVEL=1 timeout -k 10 300 python3 tools/e2e_laps.py C4 2>&1 | grep -E "solve ms|upload Ind|solve create|solve structure "
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_5.json 2> gpurun_out/r4_bench_5.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4_bench_5.json')); e=d['end_to_end']; print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], e['ms'], e['repeat_ms'], e['ms_structure'], e['ms_upload'], e['ms_pgd']); s=d['secondary_config']; print(s['value'], s['roofline']['frac'], s['end_to_end']['ms'], s['end_to_end']['repeat_ms'])"
