#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r5g
timeout -k 10 1000 python -m pytest tests/test_gpu_pair.py tests/test_gpu_train_graph.py tests/test_gpu_rl_graph.py tests/test_detector.py tests/test_gpu_bench_config.py tests/test_gpu_train_sizes.py tests/test_gpu_dp.py tests/test_gpu_parity.py tests/test_gpu_backward.py tests/test_abi_and_host.py -x -q -m gpu > gpurun_out/r5g/tests.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a gpurun_out/r5g/tests.log
grep -v "^  File\|^W2026\|^I2026\|^\[W" gpurun_out/r5g/tests.log | tail -40
[ $rc -eq 0 ] || exit $rc
bash tools/r5_ab.sh
bash tools/r5_ab2.sh
for P in 0; do ISC_PAIR_UNROLLS=$P timeout -k 10 300 python3 tools/profile_rl.py 6 2>&1 | grep -o "'ms_per_iter': [0-9.]*"; done
