#!/bin/bash
# usage (GPU box): tools/profile_next_rows.sh <tag> <workload>  -- rocprofv3 kernel statistics of tools/next_rows_bench.py
tag=${1:-next}; wl=${2:-C4}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python3 $GRAFT_REPO_ROOT/tools/next_rows_bench.py --workload $wl > $out/${wl}_next_rows.json 2> $out/${wl}_next_rows.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${wl}_stats -- python3 $GRAFT_REPO_ROOT/tools/next_rows_bench.py --workload $wl > $out/${wl}_under_rocprof.json 2>&1 || echo "stats failed"
find $out -name "*kernel_stats.csv" -path "*${wl}_stats*" -exec cp {} $out/${wl}_next_rows_kernel_stats.csv \;
find $out -name "*_kernel_trace.csv" -size +30M -delete
head -25 $out/${wl}_next_rows_kernel_stats.csv | cut -c1-180
