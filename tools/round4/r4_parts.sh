#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r4_tests_sharded7.log 2>&1; rc=$?; tail -12 gpurun_out/r4_tests_sharded7.log | cut -c1-300; [ $rc -ne 0 ] && exit $rc
for x in 2 1 4; do
DESC_SHARD_PARTS=$x timeout -k 10 300 python3 tools/shard_compute.py --workload C4 --world 8 --steps 8 --warmup 2 > gpurun_out/r04_shard_w8_c4_parts$x.json 2> gpurun_out/r04_shard_w8_c4_parts$x.err || { tail -5 gpurun_out/r04_shard_w8_c4_parts$x.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("gpurun_out/r04_shard_w8_c4_parts$x.json")); b = d["balance"]
print("parts=$x: per rank max: colsum %.1f sweep %.1f unpack %.1f; sum max %.1f; pieces" % (b["us_colsum"]["max"], b["us_sweep"]["max"], b["us_unpack"]["max"], d["compute_us_max_over_ranks"]), [r["pieces"] for r in d["ranks"]], d["exchange"]["reduce_scatter"])
PY
done
