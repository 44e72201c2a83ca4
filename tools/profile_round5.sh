# Round-5 profile set: the default bench workload (stats + PMC passes + bench line, tools/profile_round.sh), then kernel
# traces of the one-image beam-5 search (36 and 196 regions), the RL training iteration (B=512), the XE training
# iteration at B=128+80 from its HIP graph in both forms (two branches = the default inside graphs; merged step chain)
# with a launch-by-launch listing of one iteration each, and the B=128 greedy roll-out.
#   usage (on the GPU box):  bash tools/profile_round5.sh r05_a
R=$GRAFT_REPO_ROOT
TAG=${1:-r05_a}
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
export ISC_REGIONS=196
trace beam196 $R/tools/profile_beam.py 10 && cp $(newest $R/gpurun_out/prof_${TAG}_beam196) $S/${TAG}_kernel_stats_beam5_single_image_r196.csv
unset ISC_REGIONS
trace rl $R/tools/profile_rl.py 4 512 && cp $(newest $R/gpurun_out/prof_${TAG}_rl) $S/${TAG}_kernel_stats_rl_B512.csv
export ISC_PAIR=0
trace xeg $R/tools/profile_xe_graph.py 6 128 && cp $(newest $R/gpurun_out/prof_${TAG}_xeg) $S/${TAG}_kernel_stats_xe_graph.csv
python3 $R/tools/xe_graph_trace_summary.py $R/gpurun_out/prof_${TAG}_xeg $S/${TAG}_xe_graph_iteration_trace.txt > $S/${TAG}_xe_graph_summary.txt 2>&1 || true
export ISC_PAIR=1
trace xegm $R/tools/profile_xe_graph.py 6 128 && cp $(newest $R/gpurun_out/prof_${TAG}_xegm) $S/${TAG}_kernel_stats_xe_graph_merged.csv
python3 $R/tools/xe_graph_trace_summary.py $R/gpurun_out/prof_${TAG}_xegm $S/${TAG}_xe_graph_merged_iteration_trace.txt > $S/${TAG}_xe_graph_merged_summary.txt 2>&1 || true
unset ISC_PAIR
trace b128 $R/bench.py --batch 128 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-kernel-timing && cp $(newest $R/gpurun_out/prof_${TAG}_b128) $S/${TAG}_kernel_stats_b128.csv
for n in beam beam196; do grep -v "^W20\|^I20\|^E20" $R/gpurun_out/prof_${TAG}_$n.log | tail -n 1; done > $S/${TAG}_beam5_timing.txt
grep -v "^W20\|^I20\|^E20" $R/gpurun_out/prof_${TAG}_rl.log | tail -n 1 > $S/${TAG}_rl_timing.txt
# un-profiled timings of the training iteration in every form (A/B on this box)
( cd $R && bash tools/r5_ab.sh && bash tools/r5_ab2.sh ) > $S/${TAG}_xe_iteration_forms.txt 2>&1
# the traces themselves are large: only the summaries travel back
for n in beam beam196 rl xeg xegm b128; do rm -rf $R/gpurun_out/prof_${TAG}_$n; done
rm -rf $O/stats $O/fetch $O/write $O/mfma $O/clk
ls -la $S
