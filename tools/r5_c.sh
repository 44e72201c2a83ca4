#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5c
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rows.py tests/test_detector.py tests/test_gpu_rl_graph.py -x -q -m gpu > gpurun_out/r5c/tests.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a gpurun_out/r5c/tests.log
grep -v "^  File\|^W2026\|^I2026\|^\[W" gpurun_out/r5c/tests.log | tail -45
exit $rc
