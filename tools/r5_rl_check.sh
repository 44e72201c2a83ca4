#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests/test_gpu_rl_graph.py tests/test_detector.py tests/test_gpu_dp.py tests/test_gpu_train_graph.py tests/test_gpu_bench_config.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 || exit 1
for i in 1 2; do timeout -k 10 300 python3 tools/profile_rl.py 30 2>&1 | grep -o "'ms_per_iter': [0-9.]*\|'cider_ms_per_iter': [0-9.]*"; done
