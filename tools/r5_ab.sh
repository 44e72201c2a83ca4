#!/bin/bash
# A/B: merged step chain vs one chain per unroll on two streams, same box
for B in 128 256 512 1024; do
  for P in 1 0; do
    ISC_PAIR=$P timeout -k 10 300 python3 tools/profile_xe_graph.py 30 $B 2>&1 | grep "graph ms"
  done
done
