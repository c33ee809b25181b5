#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r4_tests_parity6.log 2>&1; echo "parity+sharded rc=$?"; tail -4 gpurun_out/r4_tests_parity6.log
LAPS_LEVEL=1 timeout -k 10 300 python3 tools/e2e_laps.py C4 > gpurun_out/r04_e2e_laps_c4_c.txt 2>&1
tail -34 gpurun_out/r04_e2e_laps_c4_c.txt
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_4.json 2> gpurun_out/r4_bench_4.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4_bench_4.json')); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['end_to_end']); print(d['secondary_config']['end_to_end'])"
