#!/bin/bash
# round 4: k_fill_cycles with the sample keys computed in a pass of their own; bit-exact structure tests, then the kernel's time at C4 / C5 / C2
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "structure or exact or device or kat or tie or golden" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for wl in C4 C5 C2; do
  rm -rf /tmp/fprof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fprof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-convergence > /tmp/f.log 2>&1
  echo "$wl: $(python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/fprof | grep -i "k_fill_cycles\|k_layout_node_dev" | tr '\n' ' ')"
  grep -o '"end_to_end": {"ms": [0-9.]*' /tmp/f.log; grep -o '"repeat_ms": [0-9.]*' /tmp/f.log
done
