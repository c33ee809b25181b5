#!/bin/bash
# round 4: next piece's descriptor / cycle range prefetched + the piece-switch barrier moved behind the row loads (DESC_PIECE_PREFETCH=1 build) vs production
DESC_AMD_LIB=$GRAFT_REPO_ROOT/tools/probes/libdesc_amd_prefetch.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -m gpu > gpurun_out/r4_prefetch_tests.log 2>&1; echo "tests(variant lib) rc=$?"; tail -2 gpurun_out/r4_prefetch_tests.log
bash tools/lib_ab.sh tools/probes/libdesc_amd_prefetch.so - C4 C2 C5 C3
