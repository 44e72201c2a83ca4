#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5d
timeout -k 10 900 python -m pytest tests/test_gpu_rows.py tests/test_gpu_determinism.py tests/test_abi_and_host.py -x -q -m gpu > gpurun_out/r5d/tests.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a gpurun_out/r5d/tests.log
grep -v "^  File\|^W2026\|^I2026\|^\[W" gpurun_out/r5d/tests.log | tail -30
exit $rc
