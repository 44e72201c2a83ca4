#!/bin/bash
# round 4: few-row decode path - tests, timing
set -o pipefail
mkdir -p gpurun_out/r4a
timeout -k 10 900 python -m pytest tests/test_gpu_rows.py -x -q -m gpu > gpurun_out/r4a/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r4a/tests.log
tail -15 gpurun_out/r4a/tests.log
ISC_HIP_LIB=tools/_lab/stamp/libinsenticap_hip_stamp.so timeout -k 5 200 python tools/stamp_step.py > gpurun_out/r4a/stamps.log 2>&1; grep -v amdgpu.ids gpurun_out/r4a/stamps.log
timeout -k 10 300 python tools/rows_lab.py --reps 40 > gpurun_out/r4a/lab.log 2>&1; echo "lab rc=$?"
cat gpurun_out/r4a/lab.log | tail -8
