#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4g
timeout -k 10 1100 python -m pytest tests/ -q -m gpu > gpurun_out/r4g/suite_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4g/suite_tests.log
tail -6 gpurun_out/r4g/suite_tests.log
