#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5pin
timeout -k 10 900 python -m pytest "tests/test_gpu_bench_config.py::test_config4_rl_training_iteration_at_full_size_vs_reference" -x -q -m gpu > gpurun_out/r5pin/tests.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a gpurun_out/r5pin/tests.log
grep -v "^  File\|^W2026\|^I2026\|^\[W" gpurun_out/r5pin/tests.log | tail -40
exit $rc
