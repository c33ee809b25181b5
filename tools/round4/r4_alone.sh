#!/bin/bash
for wl in C4 C5; do
timeout -k 10 400 python3 tools/shard_compute.py --workload $wl --world 8 --steps 8 --warmup 2 > gpurun_out/r04_shard_w8_${wl}_final.json 2> gpurun_out/r04_shard_w8_${wl}_final.err || { tail -5 gpurun_out/r04_shard_w8_${wl}_final.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("gpurun_out/r04_shard_w8_${wl}_final.json")); b = d["balance"]
print("$wl one GPU %.1f; interleaved: colsum %.1f sweep %.1f unpack %.1f sum %.1f | alone: colsum %.1f sweep %.1f unpack %.1f sum %.1f" % (d["one_gpu"]["us_kernel_pair"], b["us_colsum"]["max"], b["us_sweep"]["max"], b["us_unpack"]["max"], d["compute_us_max_over_ranks"], b["us_colsum_alone"]["max"], b["us_sweep_alone"]["max"], b["us_unpack_alone"]["max"], d["compute_us_max_over_ranks_alone"]))
print("   alone by rank: colsum", [round(r["us_colsum_alone"], 1) for r in d["ranks"]], "sweep", [round(r["us_sweep_alone"], 1) for r in d["ranks"]], "unpack", [round(r["us_unpack_alone"], 1) for r in d["ranks"]])
PY
done
