#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for g in 0 1; do
ISC_RL_GREEDY_STREAM=$g timeout -k 10 300 python3 tools/profile_rl.py 30 2>&1 | grep -o "'ms_per_iter': [0-9.]*" | sed "s/^/greedy_stream=$g /" || exit 1
done; done
