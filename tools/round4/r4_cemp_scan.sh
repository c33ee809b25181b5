#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for bi in 2 3 6 12; do for jb in 128 196 320 500 1000; do
  rm -rf /tmp/cprof
  DESC_DEBUG_CEMP_BI=$bi DESC_DEBUG_CEMP_JB=$jb timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof -- python3 $GRAFT_REPO_ROOT/tools/cemp_probe.py > /tmp/cemp.log 2>&1
  echo "BI=$bi JB=$jb $(python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/cprof | grep k_cemp_round)"
done; done > $GRAFT_REPO_ROOT/gpurun_out/r04_cemp_tile_scan.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r04_cemp_tile_scan.txt
