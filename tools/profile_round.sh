set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_stats.log 2>&1
python3 $R/tools/h3_op_breakdown.py $R/gpurun_out/prof_stats > $R/gpurun_out/h3_op_breakdown.json
echo stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_write.log 2>&1
echo write done
python3 $R/tools/pmc_summary.py $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write 4096 > $R/gpurun_out/pmc_summary.json
cp $R/gpurun_out/pmc_summary.json $R/profiles/r01_j_pmc_summary_B4096.json
cd $R && python3 bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
echo bench done
