#!/bin/bash
# round 4: few-row decode path - stamp labs, tests, timing
set -o pipefail
mkdir -p gpurun_out/r4a
timeout -k 5 120 tools/_lab/select_stamp_lab > gpurun_out/r4a/stamp_sel.log 2>&1; cat gpurun_out/r4a/stamp_sel.log
timeout -k 5 120 tools/_lab/rows_stamp_lab 5 > gpurun_out/r4a/stamp.log 2>&1; head -3 gpurun_out/r4a/stamp.log
timeout -k 10 900 python -m pytest tests/test_gpu_rows.py -x -q -m gpu > gpurun_out/r4a/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r4a/tests.log
tail -5 gpurun_out/r4a/tests.log
timeout -k 10 300 python tools/rows_lab.py --reps 40 > gpurun_out/r4a/lab.log 2>&1; echo "lab rc=$?"
cat gpurun_out/r4a/lab.log | tail -8
