#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r4_tests_sharded3.log 2>&1; rc=$?; tail -3 gpurun_out/r4_tests_sharded3.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/shard_compute.py --workload C4 --world 8 --steps 8 --warmup 2 > gpurun_out/r04_shard_w8_c4_v2.json 2> gpurun_out/r04_shard_w8_c4_v2.err || { tail -5 gpurun_out/r04_shard_w8_c4_v2.err; exit 1; }
timeout -k 10 300 python3 tools/shard_compute.py --workload C5 --world 8 --steps 8 --warmup 2 > gpurun_out/r04_shard_w8_c5_v2.json 2> gpurun_out/r04_shard_w8_c5_v2.err || { tail -5 gpurun_out/r04_shard_w8_c5_v2.err; exit 1; }
DESC_DEBUG_JMAJOR=1 timeout -k 10 300 python3 tools/shard_compute.py --workload C5 --world 8 --steps 8 --warmup 2 > gpurun_out/r04_shard_w8_c5_v2_jmajor1.json 2> gpurun_out/r04_shard_w8_c5_v2_jmajor1.err || { tail -5 gpurun_out/r04_shard_w8_c5_v2_jmajor1.err; exit 1; }
python3 - <<'PY'
import json
for tag in ("c4_v2", "c5_v2", "c5_v2_jmajor1"):
    d = json.load(open("gpurun_out/r04_shard_w8_%s.json" % tag)); b = d["balance"]
    print(tag, "one GPU pair %.1f us; per rank max: colsum %.1f sweep %.1f unpack %.1f; sum max %.1f" % (d["one_gpu"]["us_kernel_pair"], b["us_colsum"]["max"], b["us_sweep"]["max"], b["us_unpack"]["max"], d["compute_us_max_over_ranks"]))
    print("   colsum by rank", [round(r["us_colsum"], 1) for r in d["ranks"]], "sweep", [round(r["us_sweep"], 1) for r in d["ranks"]])
PY
bash tools/lib_ab.sh tools/probes/libdesc_amd_unordered.so - C4 C2 > gpurun_out/r04_ab_colsum_ordered.txt 2>&1
bash tools/lib_ab.sh tools/probes/libdesc_amd_r3desc.so - C4 >> gpurun_out/r04_ab_colsum_ordered.txt 2>&1
cat gpurun_out/r04_ab_colsum_ordered.txt
