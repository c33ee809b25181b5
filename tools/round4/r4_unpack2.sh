#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for t in 1 0; do
rm -rf /tmp/uprof
DESC_DEBUG_UNPACK_TILES=$t timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/uprof -- python3 $GRAFT_REPO_ROOT/tools/shard_compute.py --workload C4 --world 8 --steps 8 --warmup 2 > /tmp/u.json 2>/tmp/u.err
echo "DESC_DEBUG_UNPACK_TILES=$t"; python3 $GRAFT_REPO_ROOT/tools/kstats.py /tmp/uprof | grep -E "unpack|colsum|sweep_band" | cut -c1-140
done > $GRAFT_REPO_ROOT/gpurun_out/r04_shard_w8_c4_rocprof.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r04_shard_w8_c4_rocprof.txt
