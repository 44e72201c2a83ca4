set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03_g}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${TAG}_xeg
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_xeg -- python3 $R/tools/profile_xe_graph.py 6 128 > $R/gpurun_out/prof_${TAG}_xeg.log 2>&1
tail -1 $R/gpurun_out/prof_${TAG}_xeg.log
python3 $R/tools/xe_graph_trace_summary.py $R/gpurun_out/prof_${TAG}_xeg
