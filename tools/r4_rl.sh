#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4e
timeout -k 10 900 python -m pytest tests/test_gpu_rl_graph.py tests/test_detector.py -x -q -m gpu > gpurun_out/r4e/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r4e/tests.log
tail -30 gpurun_out/r4e/tests.log
