#!/bin/bash
# round 4, first GPU call: the new parity tests, what the box grants in CPUs, one default bench line
set -o pipefail
mkdir -p gpurun_out
{
echo "nproc $(nproc)  affinity $(python3 -c 'import os;print(len(os.sched_getaffinity(0)))')"
cat /sys/fs/cgroup/cpu.max 2>/dev/null || echo "no cgroup v2 cpu.max"
cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null || echo "no cgroup v1 quota"
python3 -c "from oracle import oracle as O; print(O.granted_cpus())"
} > gpurun_out/r4_cpus.txt 2>&1
timeout -k 10 900 python3 -m pytest tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r4_tests_sharded.log 2>&1; echo "sharded rc=$?" >> gpurun_out/r4_tests_sharded.log
tail -3 gpurun_out/r4_tests_sharded.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k oracle_parity -s > gpurun_out/r4_tests_oracle_fullsize.log 2>&1; echo "fullsize rc=$?" >> gpurun_out/r4_tests_oracle_fullsize.log
tail -5 gpurun_out/r4_tests_oracle_fullsize.log
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_first.json 2> gpurun_out/r4_bench_first.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r4_bench_first.json
