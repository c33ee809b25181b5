#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r4_tests_sharded6.log 2>&1; rc=$?; tail -3 gpurun_out/r4_tests_sharded6.log; [ $rc -ne 0 ] && exit $rc
for t in 1 0; do
DESC_DEBUG_UNPACK_TILES=$t timeout -k 10 300 python3 tools/shard_compute.py --workload C4 --world 8 --steps 8 --warmup 2 > gpurun_out/r04_shard_w8_c4_unpack$t.json 2> gpurun_out/r04_shard_w8_c4_unpack$t.err || { tail -5 gpurun_out/r04_shard_w8_c4_unpack$t.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("gpurun_out/r04_shard_w8_c4_unpack$t.json")); b = d["balance"]
print("tiles=$t: per rank max: colsum %.1f sweep %.1f unpack %.1f; sum max %.1f; unpack by rank" % (b["us_colsum"]["max"], b["us_sweep"]["max"], b["us_unpack"]["max"], d["compute_us_max_over_ranks"]), [round(r["us_unpack"], 1) for r in d["ranks"]])
PY
done
