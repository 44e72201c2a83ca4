#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5rl
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rl -- python3 tools/profile_rl.py 8 > $OUT/rl.log 2>&1; echo "prof rc=$?"
grep -v "^W2026\|^I2026\|^E2026" $OUT/rl.log | tail -1 | cut -c1-200
python3 tools/trace_busy.py $OUT/rl 1 > $OUT/rl_busy.txt 2>&1
cat $OUT/rl_busy.txt
cp $(ls $OUT/rl/*/*kernel_stats.csv | head -1) $OUT/rl_kernel_stats.csv
rm -rf $OUT/rl
