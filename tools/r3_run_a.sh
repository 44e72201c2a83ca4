set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train_sizes.py tests/test_gpu_backward.py tests/test_gpu_dp.py tests/test_bench_launcher.py -x -q -m "gpu or not gpu" > gpurun_out/a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/a_tests.log
tail -5 gpurun_out/a_tests.log
python bench.py --gpus 2 > gpurun_out/a_bench_gpus2.log 2>&1; echo "gpus2 rc=$?" >> gpurun_out/a_bench_gpus2.log
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/a_bench_torchrun1.json 2> gpurun_out/a_bench_torchrun1.err
echo "torchrun rc=$?"
tail -c 3000 gpurun_out/a_bench_torchrun1.json
