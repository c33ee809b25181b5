#!/bin/bash
# round 4: CEMP tile kernel with the per-edge set-up in scalar registers (scalar loads): tests, then kernel statistics of tools/cemp_probe.py
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_cemp.py tests/test_gpu_fullsize_next_rows.py -x -q -m gpu > gpurun_out/r4_tests_cemp2.log 2>&1; echo "cemp tests rc=$?"; tail -3 gpurun_out/r4_tests_cemp2.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof -- python3 $GRAFT_REPO_ROOT/tools/cemp_probe.py > /tmp/cemp_1.log 2>&1
tail -2 /tmp/cemp_1.log
python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/cprof | grep -i "cemp\|codeg\|bitmaps\|rank"
