#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4f
for i in 1 2; do timeout -k 10 300 python3 tools/profile_xe_graph.py 30 2>&1 | grep "graph ms"; done
timeout -k 10 300 python3 tools/profile_xe_graph.py 10 512 2>&1 | grep "graph ms"
timeout -k 10 300 python tools/rows_lab.py --reps 30 --rows-only --graphs-only 2>&1 | grep "^rows"
