"""Random-shape sweep over the GEMM entry points (linear / grouped linear / LSTM cell / vocabulary projection forward;
the backward NN / TN contractions) with a split-f16 engine forced per case - the large kernels (mode 2), the skinny
kernel with 32 x 32 or 64 x 64 tiles (modes 3 / 4), or auto inside a weights scope - each case against an fp64 reference: ragged M, K-segments, partial
column tiles, grouped problems, producer-written planes, accumulate.

    python tools/fuzz_gemm_paths.py [seed] [cases]        (on the MI355X box; tests/test_gpu_fuzz.py runs a short one)
"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from insenticap_model_amd import ops  # noqa: E402

def run(seed=0, cases=120, verbose=True):
    dev = torch.device('cuda:0')
    random.seed(seed)
    g = torch.Generator().manual_seed(1)
    def R(*s, scale=1.0): return ((torch.rand(*s, generator=g) * 2 - 1) * scale)
    bad = 0
    for it in range(cases):
        kind = random.choice(['linear', 'lstm', 'vocab', 'linear3', 'nn', 'tn'])
        M = random.choice([1, 5, 31, 33, 127, 129, 255, 257, 700, 1023, 2050, random.randint(1, 3000)])
        nseg = random.randint(1, 3)
        Ks = [32 * random.randint(1, 16) for _ in range(nseg)]
        mode = random.choice([2, 3, 4, 1])
        ops.set_h3_mode(mode)
        scope = ops.h3_weights_scope(dev)
        scope.__enter__()
        try:
            if kind in ('linear', 'linear3'):
                nprob = 1 if kind == 'linear' else random.randint(2, 3)
                probs, refs, outs, keep = [], [], [], []
                xs = [R(M, k) for k in Ks]
                for _ in range(nprob):
                    N = 4 * random.randint(1, 400)
                    ws = [R(N, k, scale=k ** -0.5) for k in Ks]
                    b = R(N)
                    relu = random.random() < 0.5
                    ref = sum(x.double() @ w.double().t() for x, w in zip(xs, ws)) + b.double()
                    if relu: ref = torch.relu(ref)
                    o = torch.empty(M, N, device=dev)
                    dseg = [(x.to(dev), w.to(dev)) for x, w in zip(xs, ws)]
                    db = b.to(dev)
                    keep.append((dseg, db))
                    probs.append(ops.linear_problem(dseg, o, db, relu=relu))
                    refs.append(ref); outs.append(o)
                ops.linear_fwd(probs)
                torch.cuda.synchronize()
                for o, r in zip(outs, refs):
                    err = (o.double().cpu() - r).abs().max().item()
                    if not err < 3e-5: bad += 1; print('BAD', kind, M, Ks, err)
            elif kind == 'lstm':
                H = 32 * random.randint(1, 16)
                xs = [R(M, k) for k in Ks]
                ws = [R(4 * H, k, scale=(3 * k) ** -0.5) for k in Ks]
                b1, b2, c0 = R(4 * H), R(4 * H), R(M, H)
                z = sum(x.double() @ w.double().t() for x, w in zip(xs, ws)) + b1.double() + b2.double()
                i, f, gg, o = z.split(H, dim=1)
                c_ref = torch.sigmoid(f) * c0.double() + torch.sigmoid(i) * torch.tanh(gg)
                h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
                dseg = [(x.to(dev), w.to(dev)) for x, w in zip(xs, ws)]
                h, c = torch.empty(M, H, device=dev), torch.empty(M, H, device=dev)
                planes = torch.zeros(2, M, H, dtype=torch.float16, device=dev)
                ops.lstm_fwd(dseg, b1.to(dev), b2.to(dev), c0.to(dev), h, c, h_planes=planes)
                torch.cuda.synchronize()
                err = max((h.double().cpu() - h_ref).abs().max().item(), (c.double().cpu() - c_ref).abs().max().item())
                # planes decode back to h
                buf = planes.cpu().view(-1).float().view(M, H // 32, 2, 32)
                back = (buf[:, :, 0, :] + buf[:, :, 1, :] / 2048.0).reshape(M, H)
                perr = (back - h.cpu()).abs().max().item()
                if not (err < 3e-5 and perr < 1e-6): bad += 1; print('BAD lstm', M, H, Ks, err, perr)
            elif kind in ('nn', 'tn'):
                N = 4 * random.randint(1, 300)
                prior = R(M, N)
                if kind == 'nn':          # C[M,N] += sum_s A_s[M,K_s] W_s[K_s,N]
                    As = [R(M, k) for k in Ks]
                    Ws = [R(k, N, scale=k ** -0.5) for k in Ks]
                    ref = prior.double() + sum(a.double() @ w.double() for a, w in zip(As, Ws))
                else:                     # C[M,N] += sum_s A_s[K_s,M]^T W_s[K_s,N]; M % 4 == 0 required
                    M = max(4, M // 4 * 4)
                    prior = R(M, N)
                    As = [R(k, M) for k in Ks]
                    Ws = [R(k, N, scale=k ** -0.5) for k in Ks]
                    ref = prior.double() + sum(a.double().t() @ w.double() for a, w in zip(As, Ws))
                out = prior.clone().to(dev)
                lay = ops.NN if kind == 'nn' else ops.TN
                ops.gemm_bwd([ops.gemm_problem([(a.to(dev), w.to(dev)) for a, w in zip(As, Ws)], out, lay,
                                               accumulate=True)], lay)
                torch.cuda.synchronize()
                err = (out.double().cpu() - ref).abs().max().item()
                if not err < 5e-5: bad += 1; print('BAD', kind, mode, M, N, Ks, err)
            else:
                V = random.choice([1000, 9487, 10000, 130, 4 * random.randint(40, 3000)])
                K = Ks[0]
                h, W, b = R(M, K), R(V, K, scale=4 * K ** -0.5), R(V)
                ref = h.double() @ W.double().t() + b.double()
                nt = (V + 127) // 128
                pm, ps = torch.empty(M, nt, device=dev), torch.empty(M, nt, device=dev)
                pi = torch.empty(M, nt, device=dev, dtype=torch.int32)
                lg = torch.empty(M, V, device=dev)
                ops.vocab_fwd(h.to(dev), W.to(dev), b.to(dev), pm, ps, pi, lg)
                torch.cuda.synchronize()
                err = (lg.double().cpu() - ref).abs().max().item()
                mx = pm.max(1).values
                lse = mx + torch.log((ps * torch.exp(pm - mx[:, None])).sum(1))
                lerr = (lse.double().cpu() - torch.logsumexp(ref, 1)).abs().max().item()
                arg = pi.gather(1, pm.argmax(1)[:, None]).squeeze(1).long()
                ok = torch.equal(arg, lg.argmax(1))
                if not (err < 5e-5 and lerr < 5e-5 and ok): bad += 1; print('BAD vocab', M, V, K, err, lerr, ok)
        except Exception as e:
            bad += 1; print('EXC', kind, M, Ks, repr(e)[:200])
        finally:
            scope.__exit__(None, None, None)
    ops.set_h3_mode(1)
    if verbose:
        print('cases done, bad =', bad, 'large / skinny split-f16 launches', ops._lib.load().isc_h3_launches(),
              ops._lib.load().isc_h3s_launches())
    return bad


if __name__ == '__main__':
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 120) else 0)
