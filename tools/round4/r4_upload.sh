#!/bin/bash
# round 4: the rotation upload on its own stream (DESC_UPLOAD_ASYNC, default on) vs on the null stream (=0): laps of desc_pgd_solve at C4 and C2, then the upload tests
for a in 1 0 1 0; do
  echo "DESC_UPLOAD_ASYNC=$a C4"; DESC_UPLOAD_ASYNC=$a timeout -k 10 300 python3 tools/e2e_laps.py C4 2>&1 | grep -E "solve ms|solve (structure|create|run)|setup_node upload|upload Ind" | tail -12
done
for a in 1 0; do
  echo "DESC_UPLOAD_ASYNC=$a C2"; DESC_UPLOAD_ASYNC=$a timeout -k 10 300 python3 tools/e2e_laps.py C2 2>&1 | grep -E "solve ms" | tail -2
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
