#!/bin/bash
# usage (GPU box): tools/ablate_bench.sh <workload> <masks...>  -- kernel-pair time of the bench with the diagnostic libraries
wl=$1; shift
for n in 0 "$@"; do
  lib=$GRAFT_REPO_ROOT/desc_amd/libdesc_amd.so; [ $n != 0 ] && lib=$GRAFT_REPO_ROOT/tools/probes/libdesc_amd_abl$n.so
  DESC_AMD_LIB=$lib timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-convergence 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ablate %3s  %s  kernel pair %.4f ms  ms/step %.4f' % ('$n', '$wl', d['roofline']['kernel_ms'], d['ms_per_step']))"
done
