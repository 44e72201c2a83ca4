#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5dp
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/profile_xe_dp.py 40 2>&1 | grep "ms per iteration"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xe -- python3 tools/profile_xe_dp.py 6 > $OUT/xe.log 2>&1; echo "prof rc=$?"
python3 tools/xe_graph_trace_summary.py $OUT/xe $OUT/xe_iteration_trace.txt > $OUT/xe_summary.txt 2>&1
head -4 $OUT/xe_summary.txt
rm -rf $OUT/xe
