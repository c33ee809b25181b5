#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "parts" 2>&1 | tail -3
timeout -k 10 700 python3 tools/fuzz_parity.py --seconds 600 --seed 4242 > gpurun_out/r04_fuzz_4242.txt 2>&1; echo "fuzz rc=$?"; tail -4 gpurun_out/r04_fuzz_4242.txt | cut -c1-400
