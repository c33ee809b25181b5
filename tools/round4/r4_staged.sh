#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4_tests_parity8.log 2>&1; echo "parity rc=$?"; tail -4 gpurun_out/r4_tests_parity8.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for st in 1 0; do
rm -rf /tmp/sprof
DESC_DEBUG_STAGED_LAYOUT=$st timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sprof -- python3 $GRAFT_REPO_ROOT/tools/e2e_laps.py C4 > /tmp/s.log 2>&1
echo "DESC_DEBUG_STAGED_LAYOUT=$st"; grep "solve ms" /tmp/s.log; python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/sprof | grep -E "layout|permute|fill_cycles|node_seg" | cut -c1-150
done > $GRAFT_REPO_ROOT/gpurun_out/r04_staged_layout.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r04_staged_layout.txt
