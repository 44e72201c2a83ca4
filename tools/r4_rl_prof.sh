#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4d
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/profile_rl.py 6 > $OUT/plain.log 2>&1; tail -1 $OUT/plain.log
