#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4f
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xe -- python3 tools/profile_xe_graph.py > $OUT/xe.log 2>&1; echo "prof rc=$?"
grep -v "^W2026\|^I2026\|^E2026" $OUT/xe.log | tail -5
