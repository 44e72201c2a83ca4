# round 3: kernel-trace statistics of the XE training iteration at B=128 (+80) and B=1024 (+80), and of B=128 greedy roll-outs
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
for cfg in "xe 6 128" "xe_B1024 4 1024"; do
  set -- $cfg
  rm -rf $R/gpurun_out/prof_${TAG}_$1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$1 -- python3 $R/tools/profile_xe.py $2 $3 > $R/gpurun_out/prof_${TAG}_$1.log 2>&1
  tail -1 $R/gpurun_out/prof_${TAG}_$1.log
done
rm -rf $R/gpurun_out/prof_${TAG}_b128
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_b128 -- python3 $R/bench.py --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-kernel-timing > $R/gpurun_out/prof_${TAG}_b128.log 2>&1
tail -c 300 $R/gpurun_out/prof_${TAG}_b128.log
