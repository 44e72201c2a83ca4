cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_detector.py tests/test_gpu_dp.py tests/test_gpu_bench_config.py -q -m gpu > gpurun_out/rl_tests.log 2>&1
echo "rc=$?"; tail -4 gpurun_out/rl_tests.log
timeout -k 10 300 python tools/profile_rl.py 5 512 2>&1 | tail -1
