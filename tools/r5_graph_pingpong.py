#!/usr/bin/env python3
"""Does hipGraphLaunch of an exec wait (on the host) for that exec's previous launch?  N replays of ONE graph of ~400 small
kernels back to back vs two captures of the same work alternated."""
import time, torch
dev = torch.device('cuda:0')
x = torch.zeros(1 << 20, device=dev)
def work():
    for _ in range(400):
        x.add_(1.0)
st = torch.cuda.Stream()
graphs = []
with torch.cuda.stream(st):
    work(); torch.cuda.synchronize()
    for _ in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            work()
        graphs.append(g)
def run(gs, n=60):
    with torch.cuda.stream(st):
        for g in gs: g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            gs[i % len(gs)].replay()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, t_host / n * 1e3
for name, gs in (('one exec', graphs[:1]), ('two execs alternated', graphs), ('one exec', graphs[:1]), ('two execs alternated', graphs)):
    ms, host = run(gs)
    print('%-22s %.3f ms per replay (host enqueue %.3f ms)' % (name, ms, host))
with torch.cuda.stream(st):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): work()
    torch.cuda.synchronize()
print('eager                  %.3f ms per 400 launches' % ((time.perf_counter() - t0) / 20 * 1e3))
