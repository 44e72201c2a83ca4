#!/usr/bin/env python3
"""RL training iterations (BASELINE configs[4]: Detector.forward, B=512) for rocprofv3 --kernel-trace --stats.
    python tools/profile_rl.py [iterations [batch]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.load_product()
dev = torch.device('cuda:0')
print(bench.bench_rl(dev, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 3, B=int(sys.argv[2]) if len(sys.argv) > 2 else 512))
