#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -q -m gpu --deselect tests/test_gpu_fullsize.py --deselect tests/test_gpu_fullsize_next_rows.py -x > gpurun_out/r4_tests_all7.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 gpurun_out/r4_tests_all7.log | cut -c1-300
timeout -k 10 300 python3 -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "C1" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for sm in 1 0; do
rm -rf /tmp/c1prof
DESC_DEBUG_SMALL=$sm timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c1prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload C1 --steps 200 --warmup 20 --no-cpu-baseline --no-convergence > /tmp/c1.json 2>/dev/null
echo "DESC_DEBUG_SMALL=$sm"; python3 -c "
import json; d=json.load(open('/tmp/c1.json')); print('ms_per_step %.5f kernel_ms %.5f value %.0f e2e %.2f repeat %.2f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['end_to_end']['ms'], d['end_to_end']['repeat_ms']))"
python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/c1prof | grep -E "sweep|colsum" | cut -c1-140
done > $GRAFT_REPO_ROOT/gpurun_out/r04_c1_small_sweep.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r04_c1_small_sweep.txt
