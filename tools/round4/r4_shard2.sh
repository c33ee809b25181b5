#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r4_tests_sharded2.log 2>&1; rc=$?; tail -3 gpurun_out/r4_tests_sharded2.log; [ $rc -ne 0 ] && exit $rc
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 tools/shard_compute.py --workload C4 --world 8 --steps 8 --warmup 2 > gpurun_out/r04_shard_w8_c4_$tag.json 2> gpurun_out/r04_shard_w8_c4_$tag.err || { echo "$tag failed"; tail -5 gpurun_out/r04_shard_w8_c4_$tag.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r04_shard_w8_c4_$tag.json"))
b=d["balance"]
print("$tag", "colsum %.1f sweep %.1f unpack %.1f (max over ranks), pieces %s, compute max %.1f" % (b["us_colsum"]["max"], b["us_sweep"]["max"], b["us_unpack"]["max"], [r["pieces"] for r in d["ranks"]][:3], d["compute_us_max_over_ranks"]))
PY
}
run default DESC_X=0 && run jmajor0 DESC_DEBUG_JMAJOR=0 && run jb196 DESC_DEBUG_JBLOCK=196 && run jb1000 DESC_DEBUG_JBLOCK=1000 && run jb2500 DESC_DEBUG_JBLOCK=2500 && run tail0 DESC_DEBUG_TAIL=0 && run aff64 DESC_DEBUG_AFFINITY=64
