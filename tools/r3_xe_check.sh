set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_sizes.py tests/test_gpu_parity.py tests/test_gpu_dp.py tests/test_detector.py -x -q -m gpu > gpurun_out/xe_check.log 2>&1 || { tail -40 gpurun_out/xe_check.log; exit 1; }
tail -3 gpurun_out/xe_check.log
bash tools/r3_quick_bench.sh --no-cpu-baseline --no-kernel-timing --steps 4
