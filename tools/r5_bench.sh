#!/bin/bash
# the default bench line, then the whole bench under a one-rank RCCL group (the data-parallel code paths)
set -o pipefail
mkdir -p gpurun_out/r5bench
timeout -k 10 900 python bench.py > gpurun_out/r5bench/bench.json 2> gpurun_out/r5bench/bench.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r5bench/bench.json; echo
grep "bench " gpurun_out/r5bench/bench.err | tail -30
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r5bench/bench_rank1.json 2> gpurun_out/r5bench/bench_rank1.err
echo "torchrun rc=$?"
tail -c 1200 gpurun_out/r5bench/bench_rank1.json; echo
