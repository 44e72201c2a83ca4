#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4b
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -- python3 tools/rows_lab.py --profile --rows-only > $OUT/eager.log 2>&1; echo "prof rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/graphs -- python3 tools/rows_lab.py --graphs-only --rows-only > $OUT/graphs.log 2>&1; echo "prof rc=$?"
grep -h "^rows" $OUT/eager.log $OUT/graphs.log
