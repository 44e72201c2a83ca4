#!/usr/bin/env python3
"""Large-M GEMM shapes of the B=4096 greedy roll-out, with torch.mm (rocBLAS/hipBLASLt fp32) as a yardstick."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from insenticap_model_amd import ops
from gemm_bench import timeit, dev

PEAK = 157.3


def report(name, us, flops):
    tf = flops / us / 1e6
    print('%-34s %8.1f us %7.1f TF  %4.1f%%' % (name, us, tf, 100 * tf / PEAK), flush=True)


def main():
    torch.backends.cuda.matmul.allow_tf32 = False
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    V = 10000
    h, W, bias = torch.randn(B, 512, device=dev), torch.randn(V, 512, device=dev) * 0.05, torch.zeros(V, device=dev)
    nt = (V + 127) // 128
    pm, ps = torch.empty(B, nt, device=dev), torch.empty(B, nt, device=dev)
    pi = torch.empty(B, nt, device=dev, dtype=torch.int32)
    fl = 2.0 * B * V * 512
    report('vocab stats-only', timeit(lambda: ops.vocab_fwd(h, W, bias, pm, ps, pi)), fl)
    lg = torch.empty(B, V, device=dev)
    report('vocab + logits', timeit(lambda: ops.vocab_fwd(h, W, bias, pm, ps, pi, lg)), fl)
    report('torch.mm vocab', timeit(lambda: torch.mm(h, W.t(), out=lg)), fl)
    for K in (1536, 1024):
        x, w = torch.randn(B, K, device=dev), torch.randn(2048, K, device=dev) * 0.02
        b = torch.zeros(2048, device=dev)
        c, ho, co = torch.randn(B, 512, device=dev), torch.empty(B, 512, device=dev), torch.empty(B, 512, device=dev)
        fl = 2.0 * B * 2048 * K
        report('lstm K=%d' % K, timeit(lambda: ops.lstm_fwd([(x, w)], b, b, c, ho, co)), fl)
        o = torch.empty(B, 2048, device=dev)
        pr = ops.linear_problem([(x, w)], o)
        report('linear N=2048 K=%d' % K, timeit(lambda: ops.linear_fwd([pr])), fl)
        report('torch.mm N=2048 K=%d' % K, timeit(lambda: torch.mm(x, w.t(), out=o)), fl)
    M = B * 36
    a, w, o = torch.randn(M, 2048, device=dev), torch.randn(512, 2048, device=dev) * 0.02, torch.empty(M, 512, device=dev)
    pr = ops.linear_problem([(a, w)], o)
    fl = 2.0 * M * 512 * 2048
    report('att_embed %dx512x2048' % M, timeit(lambda: ops.linear_fwd([pr]), reps=10), fl)
    report('torch.mm same', timeit(lambda: torch.mm(a, w.t(), out=o), reps=10), fl)
    for (N, K) in ((512, 512), (512, 1024)):
        a, w, o = torch.randn(B, K, device=dev), torch.randn(N, K, device=dev), torch.empty(B, N, device=dev)
        pr = ops.linear_problem([(a, w)], o)
        fl = 2.0 * B * N * K
        report('linear %dx%dx%d' % (B, N, K), timeit(lambda: ops.linear_fwd([pr])), fl)
        report('torch.mm same', timeit(lambda: torch.mm(a, w.t(), out=o)), fl)


if __name__ == '__main__':
    main()
