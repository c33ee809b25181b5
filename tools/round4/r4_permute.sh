#!/bin/bash
# round 4: k_permute_segments with four segments in flight per wave: parity tests, then the kernel's time at C4 / C2 / C5
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for wl in C4 C2 C5; do
  rm -rf /tmp/fprof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fprof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-convergence > /tmp/f.log 2>&1
  echo "$wl: $(python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/fprof | grep -i "k_permute_segments\|k_init_node\|k_objective" | tr '\n' ' ')"
done
