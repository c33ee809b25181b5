#!/bin/bash
mkdir -p gpurun_out
sed -i 's/os.environ\["DESC_DEBUG_TIMING"\] = "1"/os.environ["DESC_DEBUG_TIMING"] = os.environ.get("LAPS_LEVEL", "1")/' tools/e2e_laps.py
LAPS_LEVEL=2 timeout -k 10 300 python3 tools/e2e_laps.py C4 > gpurun_out/r04_e2e_laps_c4.txt 2>&1
LAPS_LEVEL=2 timeout -k 10 300 python3 tools/e2e_laps.py C2 > gpurun_out/r04_e2e_laps_c2.txt 2>&1
tail -75 gpurun_out/r04_e2e_laps_c4.txt
