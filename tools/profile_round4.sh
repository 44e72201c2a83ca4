# Round-4 profile set: the default bench workload (stats + PMC passes + bench line, tools/profile_round.sh), then kernel
# traces of the one-image beam-5 search, the RL training iteration (B=512), the XE training-graph iteration (B=128) and
# the B=128 greedy roll-out.  Collected under gpurun_out/prof_<tag>_set/ with the names profiles/ uses.
#   usage (on the GPU box):  bash tools/profile_round4.sh r04_a
R=$GRAFT_REPO_ROOT
TAG=${1:-r04_a}
S=$R/gpurun_out/prof_${TAG}_set
rm -rf $S; mkdir -p $S
bash $R/tools/profile_round.sh $TAG || { echo "profile_round failed"; exit 1; }
O=$R/gpurun_out/prof_$TAG
newest() { ls -t $1/*/*kernel_stats.csv | head -n 1; }
cp $(newest $O/stats) $S/${TAG}_kernel_stats_bench_greedy_B16384.csv
cp $O/pmc_summary.json $S/${TAG}_pmc_summary_B16384.json
cp $O/h3_op_breakdown.json $S/${TAG}_h3_op_breakdown.json
cp $O/bench.json $S/${TAG}_bench_greedy_B16384.json
cd /tmp && export TMPDIR=/tmp
trace() {   # name, then the program and its arguments
  n=$1; shift
  rm -rf $R/gpurun_out/prof_${TAG}_$n
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$n -- python3 "$@" > $R/gpurun_out/prof_${TAG}_$n.log 2>&1 || { echo "$n failed"; exit 1; }
  grep -v "^W20\|^I20\|^E20" $R/gpurun_out/prof_${TAG}_$n.log | tail -n 1
}
trace beam $R/tools/profile_beam.py 10 && cp $(newest $R/gpurun_out/prof_${TAG}_beam) $S/${TAG}_kernel_stats_beam5_single_image.csv
trace rl $R/tools/profile_rl.py 4 512 && cp $(newest $R/gpurun_out/prof_${TAG}_rl) $S/${TAG}_kernel_stats_rl_B512.csv
trace xeg $R/tools/profile_xe_graph.py 6 128 && cp $(newest $R/gpurun_out/prof_${TAG}_xeg) $S/${TAG}_kernel_stats_xe_graph.csv
python3 $R/tools/xe_graph_trace_summary.py $R/gpurun_out/prof_${TAG}_xeg > $S/${TAG}_xe_graph_summary.txt 2>&1 || true
trace b128 $R/bench.py --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-kernel-timing && cp $(newest $R/gpurun_out/prof_${TAG}_b128) $S/${TAG}_kernel_stats_b128.csv
grep -v "^W20\|^I20\|^E20" $R/gpurun_out/prof_${TAG}_beam.log | tail -n 3 > $S/${TAG}_beam5_timing.txt
grep -v "^W20\|^I20\|^E20" $R/gpurun_out/prof_${TAG}_rl.log | tail -n 1 > $S/${TAG}_rl_timing.txt
ls -la $S
