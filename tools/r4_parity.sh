#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4c
timeout -k 10 1100 python -m pytest tests/test_detector.py tests/test_gpu_bench_config.py tests/test_gpu_parity.py::test_beam_device_merge_equals_host_merge "tests/test_gpu_train_sizes.py::test_xe_train_iteration_b1024_v10k_vs_oracle_autograd" -q -m gpu > gpurun_out/r4c/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r4c/tests.log
tail -25 gpurun_out/r4c/tests.log
