set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r03_c_rl
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_c_rl -- python3 $R/tools/profile_rl.py 3 512 > $R/gpurun_out/prof_r03_c_rl.log 2>&1
grep "ms_per_iter" $R/gpurun_out/prof_r03_c_rl.log | tail -1
