#!/bin/bash
# the whole GPU suite, one process
set -o pipefail
mkdir -p gpurun_out/r5full
timeout -k 10 1100 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r5full/tests.log 2>&1
rc=$?
echo "tests rc=$rc" | tee -a gpurun_out/r5full/tests.log
grep -v "^  File\|^W2026\|^I2026" gpurun_out/r5full/tests.log | tail -45
exit $rc
