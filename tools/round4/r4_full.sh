#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
./tools/probes/colsum_probe 5000 1000 13 > gpurun_out/r04_colsum_probe.txt 2>&1; ./tools/probes/colsum_probe 5000 1000 12 >> gpurun_out/r04_colsum_probe.txt 2>&1; ./tools/probes/colsum_probe 10000 1000 9 >> gpurun_out/r04_colsum_probe.txt 2>&1
cat gpurun_out/r04_colsum_probe.txt
DESC_FORCE_SHARDED=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_forced_sharded_w1.json 2> gpurun_out/r04_bench_forced_sharded_w1.err; echo "forced sharded rc=$?"; tail -c 1500 gpurun_out/r04_bench_forced_sharded_w1.json; tail -3 gpurun_out/r04_bench_forced_sharded_w1.err
DESC_FORCE_SHARDED=1 DESC_DEBUG_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_bench_forced_collectives.json 2> gpurun_out/r04_bench_forced_collectives.err; echo "forced collectives rc=$?"; head -c 400 gpurun_out/r04_bench_forced_collectives.json
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4_tests_all1.log 2>&1; echo "all tests rc=$?"; tail -5 gpurun_out/r4_tests_all1.log
