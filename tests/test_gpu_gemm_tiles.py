"""Every GEMM tile shape (128x128, 64x128, 32x128 register-staged; 256x128 and 128x128 by LDS-DMA) behind the same
entry points: each must match an fp64 reference and - because every shape accumulates an output element
over k in the same order - must agree bit for bit with the others."""
import numpy as np
import pytest
import torch

from insenticap_model_amd import ops

pytestmark = pytest.mark.gpu
TILES = (0, 1, 2, 3, 4)


@pytest.fixture(autouse=True)
def _restore_tile_choice():
    yield
    ops.set_tile_override(-1)


def dev():
    return torch.device('cuda:0')


def _rand(g, *shape, scale=1.0):
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize('M,N,K1,K2', [(512, 256, 64, 0), (1000, 520, 512, 96), (777, 128, 2048, 0), (256, 1000, 32, 32)])
def test_linear_all_tiles(M, N, K1, K2):
    g = torch.Generator().manual_seed(M + N)
    x1, w1, b = _rand(g, M, K1), _rand(g, N, K1, scale=K1 ** -0.5), _rand(g, N)
    keep = (torch.rand(M, N, generator=g) > 0.5).to(torch.uint8)
    prior = _rand(g, M, N)
    ref = x1.double() @ w1.double().t() + b.double() + prior.double()
    segs = [(x1.to(dev()), w1.to(dev()))]
    if K2:
        x2, w2 = _rand(g, M, K2), _rand(g, N, K2, scale=K2 ** -0.5)
        ref = ref + x2.double() @ w2.double().t()
        segs.append((x2.to(dev()), w2.to(dev())))
    ref_pre = torch.relu(ref)
    ref_out = ref_pre * keep.double() * 2.0
    db, dkeep = b.to(dev()), keep.to(dev())
    outs = {}
    for t in TILES:
        ops.set_tile_override(t)
        out = prior.clone().to(dev())
        pre = torch.full((M, N), float('nan'), device=dev())
        ops.linear_fwd([ops.linear_problem(segs, out, db, relu=True, keep_mask=dkeep,
                                           mask_scale=2.0, out_pre=pre, accumulate=True)])
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), ref_out.float().numpy(), atol=3e-5, rtol=1e-5, err_msg='tile %d' % t)
        np.testing.assert_allclose(pre.cpu().numpy(), ref_pre.float().numpy(), atol=3e-5, rtol=1e-5, err_msg='tile %d' % t)
        outs[t] = out.cpu()
    for t in TILES[1:]:
        assert torch.equal(outs[0], outs[t]), 'tile %d differs from tile 0 in bits' % t


def test_linear_grouped_three_problems_all_tiles():
    g = torch.Generator().manual_seed(5)
    shapes = [(600, 256, 64), (300, 384, 128), (1030, 128, 32)]
    data = [(_rand(g, m, k), _rand(g, n, k, scale=k ** -0.5)) for m, n, k in shapes]
    refs = [(a.double() @ w.double().t()).float() for a, w in data]
    ddata = [(a.to(dev()), w.to(dev())) for a, w in data]        # problems hold raw pointers: keep these alive
    base = None
    for t in TILES:
        ops.set_tile_override(t)
        outs = [torch.empty(m, n, device=dev()) for m, n, _ in shapes]
        ops.linear_fwd([ops.linear_problem([aw], o) for aw, o in zip(ddata, outs)])
        torch.cuda.synchronize()
        for o, r in zip(outs, refs):
            np.testing.assert_allclose(o.cpu().numpy(), r.numpy(), atol=2e-5, rtol=1e-5, err_msg='tile %d' % t)
        if base is None:
            base = [o.cpu() for o in outs]
        else:
            assert all(torch.equal(b, o.cpu()) for b, o in zip(base, outs)), t


@pytest.mark.parametrize('M,H', [(700, 64), (1024, 512)])
def test_lstm_all_tiles(M, H):
    g = torch.Generator().manual_seed(M)
    ks = (H, 2 * H, 32)
    xs = [_rand(g, M, k) for k in ks]
    ws = [_rand(g, 4 * H, k, scale=(3 * k) ** -0.5) for k in ks]
    b_ih, b_hh, pre, c0 = _rand(g, 4 * H), _rand(g, 4 * H), _rand(g, M, 4 * H, scale=0.3), _rand(g, M, H)
    z = sum(x.double() @ w.double().t() for x, w in zip(xs, ws)) + b_ih.double() + b_hh.double() + pre.double()
    i, f, gg, o = z.split(H, dim=1)
    c_ref = torch.sigmoid(f) * c0.double() + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    gates_ref = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)], 1)
    dsegs = [(x.to(dev()), w.to(dev())) for x, w in zip(xs, ws)]
    dargs = [v.to(dev()) for v in (b_ih, b_hh, c0, pre)]
    base = None
    for t in TILES:
        ops.set_tile_override(t)
        h, c = torch.empty(M, H, device=dev()), torch.empty(M, H, device=dev())
        gates = torch.empty(M, 4 * H, device=dev())
        ops.lstm_fwd(dsegs, dargs[0], dargs[1], dargs[2], h, c, gates_out=gates, pre=dargs[3])
        torch.cuda.synchronize()
        np.testing.assert_allclose(h.cpu().numpy(), h_ref.float().numpy(), atol=2e-5, err_msg='tile %d' % t)
        np.testing.assert_allclose(c.cpu().numpy(), c_ref.float().numpy(), atol=2e-5, err_msg='tile %d' % t)
        np.testing.assert_allclose(gates.cpu().numpy(), gates_ref.float().numpy(), atol=2e-5, err_msg='tile %d' % t)
        if base is None:
            base = (h.cpu(), c.cpu())
        else:
            assert torch.equal(base[0], h.cpu()) and torch.equal(base[1], c.cpu()), t


@pytest.mark.parametrize('M,V,K', [(520, 1000, 64), (300, 10000, 512)])
def test_vocab_all_tiles(M, V, K):
    g = torch.Generator().manual_seed(V)
    h, W, bias = _rand(g, M, K), _rand(g, V, K, scale=4 * K ** -0.5), _rand(g, V)
    logits_ref = h.double() @ W.double().t() + bias.double()
    lse_ref = torch.logsumexp(logits_ref, 1)
    nt = (V + 127) // 128
    dh, dW, dbias = h.to(dev()), W.to(dev()), bias.to(dev())
    base = None
    for t in TILES:
        ops.set_tile_override(t)
        pm, ps = torch.empty(M, nt, device=dev()), torch.empty(M, nt, device=dev())
        pi = torch.empty(M, nt, device=dev(), dtype=torch.int32)
        logits = torch.empty(M, V, device=dev())
        ops.vocab_fwd(dh, dW, dbias, pm, ps, pi, logits)
        torch.cuda.synchronize()
        np.testing.assert_allclose(logits.cpu().numpy(), logits_ref.float().numpy(), atol=3e-5, rtol=1e-5, err_msg='tile %d' % t)
        mx = pm.max(1).values
        lse = mx + torch.log((ps * torch.exp(pm - mx[:, None])).sum(1))
        np.testing.assert_allclose(lse.cpu().numpy(), lse_ref.float().numpy(), atol=3e-5, rtol=1e-5, err_msg='tile %d' % t)
        col = pm.argmax(1)                                  # first tile holding the row maximum
        arg = pi.gather(1, col[:, None]).squeeze(1).long()
        ref_arg = logits.argmax(1)                          # same fp32 values -> same arg-max, ties to the lower id
        assert torch.equal(arg, ref_arg), 'tile %d' % t
        if base is None:
            base = (logits.cpu(), pm.cpu(), pi.cpu())
        else:
            assert torch.equal(base[0], logits.cpu()) and torch.equal(base[1], pm.cpu()) and torch.equal(base[2], pi.cpu()), t


def test_large_shapes_bit_identical_across_big_tiles():
    """The roll-out's own shapes at B=4096 (full grids, every CU busy, operands streaming from HBM): the
    LDS-DMA tiles must reproduce the register-staged 128x128 tile bit for bit."""
    g = torch.Generator().manual_seed(77)
    B = 4096
    x = _rand(g, B, 1536).to(dev())
    w = _rand(g, 2048, 1536, scale=1536 ** -0.5).to(dev())
    b = _rand(g, 2048).to(dev())
    c0 = _rand(g, B, 512).to(dev())
    hW = _rand(g, 10000, 512, scale=4 * 512 ** -0.5).to(dev())
    hb = _rand(g, 10000).to(dev())
    base = {}
    for t in (0, 3, 4):
        ops.set_tile_override(t)
        out = torch.empty(B, 2048, device=dev())
        ops.linear_fwd([ops.linear_problem([(x[:, :1024], w[:, :1024]), (x[:, 1024:], w[:, 1024:])], out, b, relu=True)])
        h, c = torch.empty(B, 512, device=dev()), torch.empty(B, 512, device=dev())
        ops.lstm_fwd([(x, w)], b, b, c0, h, c)
        nt = 79
        pm, ps = torch.empty(B, nt, device=dev()), torch.empty(B, nt, device=dev())
        pi = torch.empty(B, nt, device=dev(), dtype=torch.int32)
        ops.vocab_fwd(h, hW, hb, pm, ps, pi)
        torch.cuda.synchronize()
        cur = dict(lin=out.cpu(), h=h.cpu(), c=c.cpu(), pm=pm.cpu(), pi=pi.cpu())
        if not base:
            base = cur
            ref = torch.relu(x.double() @ w.double().t() + b.double()).float().cpu()
            np.testing.assert_allclose(cur['lin'].numpy(), ref.numpy(), atol=3e-5, rtol=1e-5)
        else:
            for k in base:
                assert torch.equal(base[k], cur[k]), (t, k)
