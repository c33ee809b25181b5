#!/bin/bash
# usage (GPU box): tools/ablate_pmc.sh <workload> <masks...>  -- FETCH_SIZE / WRITE_SIZE per launch of the bench with the diagnostic libraries
wl=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/abl_pmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for n in 0 "$@"; do
  lib=$GRAFT_REPO_ROOT/desc_amd/libdesc_amd.so; [ $n != 0 ] && lib=$GRAFT_REPO_ROOT/tools/probes/libdesc_amd_abl$n.so
  export DESC_AMD_LIB=$lib
  for set in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/abl_pmc_$set
    timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/abl_pmc_$set -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-convergence > /dev/null 2>&1 || echo "pmc $set failed"
  done
  echo "== ablate $n $wl"
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py ${wl}_abl$n /tmp/abl_pmc_FETCH_SIZE /tmp/abl_pmc_WRITE_SIZE --into $out/traffic_$n.json
done
