#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5a
timeout -k 10 600 python -m pytest tests/test_detector.py tests/test_gpu_bench_config.py -q -m gpu > gpurun_out/r5a/tests2.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a gpurun_out/r5a/tests2.log
grep -v "^  File\|^W2026\|^I2026" gpurun_out/r5a/tests2.log | tail -40
for i in 1 2; do timeout -k 10 300 python3 tools/profile_xe_graph.py 30 2>&1 | grep "graph ms"; done
timeout -k 10 300 python3 tools/profile_xe_graph.py 10 512 2>&1 | grep "graph ms"
timeout -k 10 300 python3 tools/profile_xe_graph.py 10 1024 2>&1 | grep "graph ms"
exit $rc
