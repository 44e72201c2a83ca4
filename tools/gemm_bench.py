#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM entry points (device time per launch from HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from insenticap_model_amd import ops

dev = torch.device('cuda:0')


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def nt(M, N, K):
    a, w, o = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
    pr = ops.linear_problem([(a, w)], o)
    return timeit(lambda: ops.linear_fwd([pr]))


def nn(M, N, K):
    a, w, o = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev), torch.empty(M, N, device=dev)
    pr = ops.gemm_problem([(a, w)], o, ops.NN)
    return timeit(lambda: ops.gemm_bwd([pr], ops.NN))


def tn(R, M, N):
    a, w, o = torch.randn(R, M, device=dev), torch.randn(R, N, device=dev), torch.empty(M, N, device=dev)
    pr = ops.gemm_problem([(a, w)], o, ops.TN)
    return timeit(lambda: ops.gemm_bwd([pr], ops.TN))


def lstm(M, H, K):
    x, w = torch.randn(M, K, device=dev), torch.randn(4 * H, K, device=dev)
    b = torch.zeros(4 * H, device=dev)
    c, ho, co = torch.randn(M, H, device=dev), torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
    return timeit(lambda: ops.lstm_fwd([(x, w)], b, b, c, ho, co))


if __name__ == '__main__':
    print('empty-ish launch (M=4,N=128,K=32): %.1f us' % nt(4, 128, 32))
    for M in (4, 128, 320, 512, 2048):
        print('M=%d' % M)
        for (N, K) in ((512, 512), (512, 1024), (512, 2048), (2048, 512), (10000, 512)):
            t = nt(M, N, K)
            print('  NT N=%5d K=%5d: %7.1f us  %6.1f TF' % (N, K, t, 2.0 * M * N * K / t / 1e6))
        t = lstm(M, 512, 1536)
        print('  LSTM 4H=2048 K=1536: %7.1f us  %6.1f TF' % (t, 2.0 * M * 2048 * 1536 / t / 1e6))
        for (N, K) in ((512, 2048), (1536, 2048), (512, 10000)):
            t = nn(M, N, K)
            print('  NN N=%5d K=%5d: %7.1f us  %6.1f TF' % (N, K, t, 2.0 * M * N * K / t / 1e6))
    for (R, M, N) in ((2560, 2048, 1536), (2560, 10000, 512), (2560, 512, 512), (128, 2048, 512)):
        t = tn(R, M, N)
        print('TN rows=%5d M=%5d N=%5d: %7.1f us  %6.1f TF' % (R, M, N, t, 2.0 * R * M * N / t / 1e6))
