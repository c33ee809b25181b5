#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4_tests_parity9.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r4_tests_parity9.log | cut -c1-300
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "oracle_parity or unsampled" > gpurun_out/r4_tests_fs9.log 2>&1; echo "fullsize rc=$?"; tail -3 gpurun_out/r4_tests_fs9.log | cut -c1-300
LAPS_LEVEL=1 timeout -k 10 300 python3 tools/e2e_laps.py C4 2>&1 | grep -E "solve ms|compaction|tables to host|solve structure |solve create"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_6.json 2> gpurun_out/r4_bench_6.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4_bench_6.json')); e=d['end_to_end']; print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], e['ms'], e['repeat_ms'], e['ms_structure'], e['ms_upload'], e['ms_pgd']); s=d['secondary_config']; print(s['value'], s['roofline']['frac'], s['end_to_end']['ms'], s['end_to_end']['repeat_ms'])"
