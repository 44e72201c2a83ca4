set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_detector.py tests/test_gpu_dp.py tests/test_gpu_backward.py tests/test_gpu_train_sizes.py tests/test_gpu_h3.py -q -m gpu > gpurun_out/c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c_tests.log
tail -30 gpurun_out/c_tests.log
