#!/bin/bash
# final profile set of round 4 (r04c): kernel stats + PMC traffic for C4 / C2 / C1, the default bench line, next rows at C4
mkdir -p gpurun_out
bash tools/profile_round.sh r04cprof C4 C2 C1 > gpurun_out/r04cprof.log 2>&1
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > gpurun_out/r04c_bench_default.json 2> gpurun_out/r04c_bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python3 tools/next_rows_bench.py --workload C4 > gpurun_out/r04c_c4_next_rows.json 2> gpurun_out/r04c_c4_next_rows.err; echo "next rows rc=$?"
ls gpurun_out/r04cprof | head -40
cat gpurun_out/r04cprof/traffic.json | head -c 1200
