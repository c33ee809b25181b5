#!/bin/bash
# round 4: the Python wrapper on native marshalling + the one-call path: tests, then the wrapper's wall time at C4
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_next_rows.py -x -q -m gpu -k "wrapper or DESC or progress or make_plots or demo or mex or marshal" > gpurun_out/r4_wrap_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_wrap_tests.log
timeout -k 10 500 python3 tools/next_rows_bench.py --workload C4 > gpurun_out/r04_c4_next_rows_wrap.json 2> gpurun_out/r04_c4_next_rows_wrap.err; echo "bench rc=$?"
python3 - <<'PY'
import json
for line in open("gpurun_out/r04_c4_next_rows_wrap.json"):
    d = json.loads(line)
    if "device_resident_problem" in d:
        r = d["device_resident_problem"]
        print({k: round(v, 1) for k, v in r.items() if k.endswith("_ms")})
    if "desc_pgd_solve" in d: print("solve", d["desc_pgd_solve"])
PY
