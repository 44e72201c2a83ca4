#!/bin/bash
# kernel trace of graph-served XE iterations (B = 128 + 80) + per-iteration summary
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5p
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xe -- python3 tools/profile_xe_graph.py ${1:-6} ${2:-128} > $OUT/xe.log 2>&1; echo "prof rc=$?"
grep -v "^W2026\|^I2026\|^E2026" $OUT/xe.log | tail -3
python3 tools/xe_graph_trace_summary.py $OUT/xe $OUT/xe_iteration_trace.txt > $OUT/xe_graph_summary.txt 2>&1
head -8 $OUT/xe_graph_summary.txt
cp $(ls $OUT/xe/*/*kernel_stats.csv | head -1) $OUT/kernel_stats_xe_graph.csv
# the trace itself is large: keep only the summaries
rm -rf $OUT/xe
