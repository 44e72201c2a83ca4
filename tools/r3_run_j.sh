cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -q -m gpu -x > gpurun_out/j_tests.log 2>&1
echo "first rc=$?"; tail -12 gpurun_out/j_tests.log
