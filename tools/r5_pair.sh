#!/bin/bash
# round 5: the merged step chain (tests/test_gpu_pair.py) + every test that trains, then iteration timings
set -o pipefail
mkdir -p gpurun_out/r5a
timeout -k 10 1000 python -m pytest tests/test_gpu_pair.py tests/test_gpu_train_graph.py tests/test_gpu_rl_graph.py tests/test_detector.py tests/test_gpu_bench_config.py tests/test_gpu_train_sizes.py tests/test_gpu_dp.py tests/test_gpu_parity.py -q -m gpu > gpurun_out/r5a/tests.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a gpurun_out/r5a/tests.log
grep -v "^  File\|^W2026\|^I2026" gpurun_out/r5a/tests.log | tail -60
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do timeout -k 10 300 python3 tools/profile_xe_graph.py 30 2>&1 | grep "graph ms"; done
timeout -k 10 300 python3 tools/profile_xe_graph.py 10 512 2>&1 | grep "graph ms"
timeout -k 10 300 python3 tools/profile_xe_graph.py 10 1024 2>&1 | grep "graph ms"
