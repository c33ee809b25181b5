#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4_tests_parity5.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r4_tests_parity5.log
LAPS_LEVEL=1 timeout -k 10 300 python3 tools/e2e_laps.py C4 > gpurun_out/r04_e2e_laps_c4_b.txt 2>&1
grep -E "solve ms|create|structure  |upload Ind|layout kernels|setup_node plan" gpurun_out/r04_e2e_laps_c4_b.txt
DESC_DEBUG_EARLY_LAYOUT=0 LAPS_LEVEL=1 timeout -k 10 300 python3 tools/e2e_laps.py C4 2>&1 | grep -E "solve ms|solve create"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/r4_bench_3.json 2> gpurun_out/r4_bench_3.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4_bench_3.json')); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['end_to_end'])"
