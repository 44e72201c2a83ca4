#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5x1024
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export ISC_PAIR=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xe -- python3 tools/profile_xe_graph.py 6 ${1:-1024} > $OUT/xe.log 2>&1; echo "prof rc=$?"
grep -v "^W2026\|^I2026\|^E2026" $OUT/xe.log | tail -2
python3 tools/xe_graph_trace_summary.py $OUT/xe $OUT/xe_iteration_trace.txt > $OUT/xe_graph_summary.txt 2>&1
head -40 $OUT/xe_graph_summary.txt
rm -rf $OUT/xe
