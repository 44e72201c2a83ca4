#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5f
timeout -k 10 900 python -m pytest tests/test_gpu_h3.py tests/test_gpu_backward.py tests/test_gpu_train_sizes.py tests/test_gpu_pair.py -x -q -m gpu > gpurun_out/r5f/tests.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a gpurun_out/r5f/tests.log
grep -v "^  File\|^W2026\|^I2026\|^\[W" gpurun_out/r5f/tests.log | tail -30
[ $rc -eq 0 ] || exit $rc
bash tools/r5_ab.sh
