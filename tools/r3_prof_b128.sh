set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r03_d_b128 $R/gpurun_out/prof_r03_d_beam
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_d_b128 -- python3 $R/bench.py --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-kernel-timing > $R/gpurun_out/prof_r03_d_b128.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_d_beam -- python3 $R/tools/profile_beam.py 10 > $R/gpurun_out/prof_r03_d_beam.log 2>&1
echo done
