#!/bin/bash
# round 4: edges in flight per wave in the CEMP tile kernel (DESC_CEMP_U builds), kernel statistics of tools/cemp_probe.py
cd /tmp && export TMPDIR=/tmp
for lib in desc_amd/libdesc_amd.so tools/probes/libdesc_amd_cempu2.so tools/probes/libdesc_amd_cempu6.so tools/probes/libdesc_amd_cempu8.so desc_amd/libdesc_amd.so; do
  rm -rf /tmp/cprof
  DESC_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof -- python3 $GRAFT_REPO_ROOT/tools/cemp_probe.py > /tmp/cemp_1.log 2>&1
  echo "$lib: $(python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/cprof | grep -i "round_ti")"
done
