# Kernel-trace statistics of the small-batch (latency-bound) regimes: greedy roll-out at B=128 and the XE training step.
# usage (on the GPU box): bash tools/profile_small.sh <tag>
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-small}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}_b128 $R/gpurun_out/prof_${TAG}_xe
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_b128 -- python3 $R/bench.py --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-kernel-timing > $R/gpurun_out/prof_${TAG}_b128.log 2>&1
echo b128 done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_xe -- python3 $R/tools/profile_xe.py > $R/gpurun_out/prof_${TAG}_xe.log 2>&1
echo xe done
