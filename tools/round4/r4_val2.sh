#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -q -m gpu --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_fullsize_next_rows.py -x > gpurun_out/r4_tests_all6.log 2>&1; rc=$?; echo "tests (without the full-size files) rc=$rc"; tail -8 gpurun_out/r4_tests_all6.log | cut -c1-300
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python3 tools/shard_compute.py --workload C4 --world 8 --steps 8 --warmup 2 > gpurun_out/r04_shard_w8_c4_v5.json 2> gpurun_out/r04_shard_w8_c4_v5.err || { tail -5 gpurun_out/r04_shard_w8_c4_v5.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("gpurun_out/r04_shard_w8_c4_v5.json")); b = d["balance"]
print("per rank max: colsum %.1f sweep %.1f unpack %.1f; sum max %.1f | alone colsum %.1f sweep %.1f unpack %.1f sum %.1f" % (b["us_colsum"]["max"], b["us_sweep"]["max"], b["us_unpack"]["max"], d["compute_us_max_over_ranks"], b["us_colsum_alone"]["max"], b["us_sweep_alone"]["max"], b["us_unpack_alone"]["max"], d["compute_us_max_over_ranks_alone"]))
PY
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fullsize_next_rows.py -q -m gpu -x > gpurun_out/r4_tests_fullsize6.log 2>&1; rc=$?; echo "full-size rc=$rc"; tail -6 gpurun_out/r4_tests_fullsize6.log | cut -c1-300
