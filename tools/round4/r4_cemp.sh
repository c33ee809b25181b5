#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_cemp.py tests/test_gpu_fullsize_next_rows.py -x -q -m gpu > gpurun_out/r4_tests_cemp.log 2>&1; echo "cemp tests rc=$?"; tail -5 gpurun_out/r4_tests_cemp.log
cd /tmp && export TMPDIR=/tmp
for t in 1 0; do
  rm -rf /tmp/cprof
  DESC_DEBUG_CEMP_TILES=$t timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof -- python3 $GRAFT_REPO_ROOT/tools/cemp_probe.py > /tmp/cemp_$t.log 2>&1
  echo "DESC_DEBUG_CEMP_TILES=$t"; tail -2 /tmp/cemp_$t.log
  python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/cprof | grep -i "cemp\|codeg\|bitmaps\|rank" 
done > $GRAFT_REPO_ROOT/gpurun_out/r04_cemp_tiles.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r04_cemp_tiles.txt
