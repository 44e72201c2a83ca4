#!/usr/bin/env python3
"""Times the few-row GEMM entry points (the decode step's launches at B <= a few hundred) per launch with HIP events:
    python tools/skinny_bench.py [--mode 3] [--rows 5,32,128,320]
mode 3 / 4 = skinny split-f16 kernel with 32 x 32 / 64 x 64 tiles, 2 = large split-f16 kernels, 0 = fp32 split-K + reduce, 1 = auto."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from insenticap_model_amd import ops


def timed(fn, reps=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', type=int, default=3)
    ap.add_argument('--rows', default='5,32,128,320')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    H, V = 512, 10000
    W1 = torch.randn(4 * H, 3 * H, generator=g).to(dev) * 0.03        # lstm weight_ih [2048, 1536]
    Whh = torch.randn(4 * H, H, generator=g).to(dev) * 0.03
    Wc = torch.randn(V, H, generator=g).to(dev) * 0.03
    bc = torch.randn(V, generator=g).to(dev)
    Wp = [torch.randn(H, H, generator=g).to(dev) * 0.03 for _ in range(3)]
    bp = [torch.randn(H, generator=g).to(dev) for _ in range(3)]
    b4 = torch.randn(4 * H, generator=g).to(dev)
    ops.set_h3_mode(a.mode)
    for M in [int(x) for x in a.rows.split(',')]:
        x = [torch.randn(M, H, generator=g).to(dev) for _ in range(3)]
        c0 = torch.randn(M, H, generator=g).to(dev)
        h, c = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
        nt = (V + 127) // 128
        pm, ps = torch.empty(M, nt, device=dev), torch.empty(M, nt, device=dev)
        pi = torch.empty(M, nt, dtype=torch.int32, device=dev)
        outs = [torch.empty(M, H, device=dev) for _ in range(3)]
        hp = torch.empty(2, M, H, dtype=torch.float16, device=dev)
        with ops.h3_weights_scope(dev):
            res = {}
            res['lstm K=1536 (fp32 A)'] = timed(lambda: ops.lstm_fwd(
                [(x[0], W1[:, 0:H]), (x[1], W1[:, H:2 * H]), (x[2], Whh)], b4, b4, c0, h, c))
            ops.lstm_fwd([(x[0], W1[:, 0:H])], b4, b4, c0, h, c, h_planes=hp)      # planes of h for the next ones
            res['lstm K=1024 (planes)'] = timed(lambda: ops.lstm_fwd(
                [(h, W1[:, 0:H], hp), (h, Whh, hp)], b4, b4, c0, x[0], c))
            res['proj 3x[512x512] (planes)'] = timed(lambda: ops.linear_fwd(
                [ops.linear_problem([(h, Wp[i], hp)], outs[i], bp[i]) for i in range(3)]))
            res['gate [512 x 1024] (fp32 A)'] = timed(lambda: ops.linear_fwd(
                [ops.linear_problem([(x[0], Wp[0]), (x[1], Wp[1])], outs[0], bp[0], accumulate=True)]))
            res['vocab K=512 (planes)'] = timed(lambda: ops.vocab_fwd(h, Wc, bc, pm, ps, pi, h_planes=hp))
        print('M=%4d mode %d  ' % (M, a.mode) + '  '.join('%s %.1f us' % kv for kv in res.items()), flush=True)


if __name__ == '__main__':
    main()
