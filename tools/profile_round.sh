#!/bin/bash
# usage (GPU box): tools/profile_round.sh <tag>   -- kernel-trace stats of the default bench (C2) and of C4,
# then the two HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) on C2.  Outputs under gpurun_out/<tag>/.
tag=${1:-prof}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c2_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-convergence > $out/c2_bench.log 2>&1 || echo "c2 stats failed"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4_stats -- python3 $GRAFT_REPO_ROOT/bench.py --workload C4 --steps 20 --warmup 5 --no-convergence > $out/c4_bench.log 2>&1 || echo "c4 stats failed"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/c2_pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-convergence > $out/c2_pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
