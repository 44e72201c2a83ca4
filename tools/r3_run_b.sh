set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -q -m gpu > gpurun_out/b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/b_tests.log
tail -30 gpurun_out/b_tests.log
