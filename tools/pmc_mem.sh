#!/bin/bash
# usage: tools/pmc_mem.sh <outdir> <program args...>  -- memory-side counters, one rocprofv3 run per
# counter set (few TCC counters per pass: the hardware refuses larger sets), each under a timeout
out=$1; shift
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE TCC_MISS_sum" \
           "WRITE_SIZE TCC_HIT_sum TCC_REQ_sum" \
           "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCP_TCC_WRITE_REQ_sum TCP_GATE_EN1_sum TD_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pass$i -- "$@" > $GRAFT_REPO_ROOT/$out/pass$i.log 2>&1 || echo "pass $i failed"
done
