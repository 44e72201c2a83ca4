#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5b
timeout -k 10 900 python -m pytest tests/test_gpu_h3.py tests/test_gpu_pair.py tests/test_gpu_train_graph.py tests/test_gpu_gemm_tiles.py -x -q -m gpu > gpurun_out/r5b/tests.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a gpurun_out/r5b/tests.log
grep -v "^  File\|^W2026\|^I2026" gpurun_out/r5b/tests.log | tail -40
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do timeout -k 10 300 python3 tools/profile_xe_graph.py 30 2>&1 | grep "graph ms"; done
timeout -k 10 300 python3 tools/profile_xe_graph.py 10 512 2>&1 | grep "graph ms"
timeout -k 10 300 python3 tools/profile_xe_graph.py 10 1024 2>&1 | grep "graph ms"
bash tools/r5_xe_prof.sh
