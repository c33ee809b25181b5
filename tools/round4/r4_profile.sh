#!/bin/bash
# round-4 profile set: kernel stats + PMC traffic for C1..C5, the default bench line, next rows
mkdir -p gpurun_out
bash tools/profile_round.sh r04prof C4 C2 C5 C3 C1 > gpurun_out/r04prof.log 2>&1
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python3 tools/next_rows_bench.py --workload C4 > gpurun_out/r04_c4_next_rows.json 2> gpurun_out/r04_c4_next_rows.err; echo "next rows rc=$?"
bash tools/profile_next_rows.sh r04next C4 > gpurun_out/r04next.log 2>&1
ls gpurun_out/r04prof | head -40
cat gpurun_out/r04prof/traffic.json | head -c 1500
