set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -q -m gpu > gpurun_out/e_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/e_tests.log
tail -8 gpurun_out/e_tests.log
timeout -k 10 300 python tools/profile_xe.py 6 128 > gpurun_out/e_xe128.log 2>&1; tail -2 gpurun_out/e_xe128.log
timeout -k 10 300 python tools/profile_xe.py 4 1024 > gpurun_out/e_xe1024.log 2>&1; tail -2 gpurun_out/e_xe1024.log
