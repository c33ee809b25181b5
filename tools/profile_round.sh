#!/bin/bash
# usage (GPU box): tools/profile_round.sh <tag> [workloads...]   -- per workload: kernel-trace stats of bench.py, then
# the two HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs: they do not fit one pass).
# Outputs under gpurun_out/<tag>/; summaries are copied into profiles/ by hand (tools/pmc_traffic.py).
tag=${1:-prof}; shift
wls=${@:-C2 C4}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in $wls; do
  steps=30; [ $wl = C2 ] && steps=100; [ $wl = C1 ] && steps=100
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${wl}_stats -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps $steps --warmup 5 --no-convergence --no-cpu-baseline > $out/${wl}_bench.log 2>&1 || echo "$wl stats failed"
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/${wl}_pmc_$set -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-convergence > $out/${wl}_pmc_$set.log 2>&1 || echo "$wl pmc $set failed"
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py $wl $out/${wl}_pmc_FETCH_SIZE $out/${wl}_pmc_WRITE_SIZE --into $out/traffic.json > $out/${wl}_traffic.log 2>&1
  find $out -name "*kernel_stats.csv" -path "*${wl}_stats*" -exec cp {} $out/${wl}_kernel_stats.csv \;
  find $out -name "*counter_collection.csv" -size +20M -delete
done
