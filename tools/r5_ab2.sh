#!/bin/bash
for B in 128 512; do for P in 1 0; do ISC_PAIR=$P timeout -k 10 300 python3 tools/r5_ab_eager.py 20 $B 2>&1 | grep "eager ms"; done; done
