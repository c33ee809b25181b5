#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python3 tools/shard_compute.py --workload C4 --world 8 > gpurun_out/r04_shard_w8_c4.json 2> gpurun_out/r04_shard_w8_c4.err; echo "c4 rc=$?"
timeout -k 10 500 python3 tools/shard_compute.py --workload C5 --world 8 > gpurun_out/r04_shard_w8_c5.json 2> gpurun_out/r04_shard_w8_c5.err; echo "c5 rc=$?"
timeout -k 10 450 python3 tools/fuzz_parity.py --seconds 400 --seed 2026 > gpurun_out/r04_fuzz_2026.txt 2>&1; echo "fuzz rc=$?"
tail -5 gpurun_out/r04_fuzz_2026.txt
