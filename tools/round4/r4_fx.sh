#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r4_tests_all2.log 2>&1; rc=$?; echo "all tests rc=$rc"; tail -15 gpurun_out/r4_tests_all2.log
bash tools/lib_ab.sh tools/probes/libdesc_amd_unordered.so - C4 C2 C5 C3 > gpurun_out/r04_ab_colsum_fixed_point.txt 2>&1
cat gpurun_out/r04_ab_colsum_fixed_point.txt
timeout -k 10 300 python3 tools/shard_compute.py --workload C4 --world 8 --steps 8 --warmup 2 > gpurun_out/r04_shard_w8_c4_v3.json 2> gpurun_out/r04_shard_w8_c4_v3.err || { tail -5 gpurun_out/r04_shard_w8_c4_v3.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_shard_w8_c4_v3.json")); b = d["balance"]
print("c4_v3 one GPU pair %.1f us; per rank max: colsum %.1f sweep %.1f unpack %.1f; sum max %.1f" % (d["one_gpu"]["us_kernel_pair"], b["us_colsum"]["max"], b["us_sweep"]["max"], b["us_unpack"]["max"], d["compute_us_max_over_ranks"]))
print("   colsum by rank", [round(r["us_colsum"], 1) for r in d["ranks"]], "sweep", [round(r["us_sweep"], 1) for r in d["ranks"]])
PY
