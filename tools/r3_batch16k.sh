# parity at the new bench batch, then the profile set at it
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_bench_config.py -x -q -m gpu -k "bench_batch" > gpurun_out/b16k_test.log 2>&1 || { tail -30 gpurun_out/b16k_test.log; exit 1; }
tail -3 gpurun_out/b16k_test.log
bash tools/profile_round.sh r03_e 16384
tail -c 1500 gpurun_out/prof_r03_e/bench.json
