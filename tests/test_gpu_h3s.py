"""Skinny split-f16 GEMM path (gemm_h3s_kernel; isc_set_h3_mode(3) / (4) force its 32 x 32 / 64 x 64 tile, auto mode
takes it for few-row launches inside a weights scope): one launch per GEMM instead of split-K slabs + a reduce kernel.  Checked against fp64 through
every epilogue (linear with bias / ReLU / accumulate / keep-mask / pre-activation copy, LSTM cell with hoisted terms
and the token table, vocabulary statistics with and without logits), on ragged shapes, mixed plane / fp32 activation
segments, grouped launches; bit-repeatable; and that auto mode selects it exactly where it should."""
import numpy as np
import pytest
import torch

from insenticap_model_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore():
    yield
    ops.set_h3_mode(1)
    ops.set_tile_override(-1)
    ops.set_gemv_rows(8)


def dev():
    return torch.device('cuda:0')


def _rand(g, *shape, scale=1.0):
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _planes(x):
    """[2, M, K] f16 buffer in the library's interleaved layout (isc_seg.A_hi): per row and 32-k block 32 hi then 32 lo."""
    M, K = x.shape
    hi = x.to(torch.float16)
    lo = ((x - hi.float()) * 2048.0).to(torch.float16)
    buf = torch.stack([hi.view(M, K // 32, 32), lo.view(M, K // 32, 32)], dim=2)       # [M, K/32, 2, 32]
    return buf.reshape(2, M, K).contiguous()


def _launches():
    return ops._lib.load().isc_h3s_launches()


@pytest.mark.parametrize('M,N,K1,K2,planes', [(5, 512, 512, 0, False), (37, 96, 64, 32, True), (128, 520, 512, 512, True),
                                              (300, 1536, 1024, 0, False), (1, 32, 32, 0, False), (80, 2048, 96, 32, False), (1000, 1100, 1536, 64, True)])
@pytest.mark.parametrize('skinny', [3, 4])
def test_linear_skinny_vs_fp64(M, N, K1, K2, planes, skinny):
    g = torch.Generator().manual_seed(M * 31 + N)
    x1, w1, b = _rand(g, M, K1), _rand(g, N, K1, scale=K1 ** -0.5), _rand(g, N)
    keep = (torch.rand(M, N, generator=g) > 0.5).to(torch.uint8)
    prior = _rand(g, M, N)
    ref = x1.double() @ w1.double().t() + b.double() + prior.double()
    dx1 = x1.to(dev())
    segs = [(dx1, w1.to(dev()), _planes(dx1)) if planes else (dx1, w1.to(dev()))]       # planes for segment 0 only:
    if K2:                                                                              # segment 1 is split in registers
        x2, w2 = _rand(g, M, K2), _rand(g, N, K2, scale=K2 ** -0.5)
        ref = ref + x2.double() @ w2.double().t()
        segs.append((x2.to(dev()), w2.to(dev())))
    ref_pre = torch.relu(ref)
    ref_out = ref_pre * keep.double() * 2.0
    outs = []
    for rep in range(2):
        ops.set_h3_mode(skinny)
        n0 = _launches()
        out = prior.clone().to(dev())
        pre = torch.full((M, N), float('nan'), device=dev())
        ops.linear_fwd([ops.linear_problem(segs, out, b.to(dev()), relu=True, keep_mask=keep.to(dev()),
                                           mask_scale=2.0, out_pre=pre, accumulate=True)])
        torch.cuda.synchronize()
        assert _launches() == n0 + 1
        np.testing.assert_allclose(out.cpu().numpy(), ref_out.float().numpy(), atol=3e-5, rtol=1e-5)
        np.testing.assert_allclose(pre.cpu().numpy(), ref_pre.float().numpy(), atol=3e-5, rtol=1e-5)
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])                                # bit-repeatable
    # fp32-accurate: no worse than the exact-fp32 MFMA tiles (few rows: keep the fused matrix-vector kernel out of it)
    ops.set_h3_mode(0)
    ops.set_gemv_rows(0)
    out0 = prior.clone().to(dev())
    pre0 = torch.empty(M, N, device=dev())
    ops.linear_fwd([ops.linear_problem(segs, out0, b.to(dev()), relu=True, keep_mask=keep.to(dev()),
                                       mask_scale=2.0, out_pre=pre0, accumulate=True)])
    torch.cuda.synchronize()
    e3 = (pre.double().cpu() - ref_pre).pow(2).mean().sqrt().item()
    e0 = (pre0.double().cpu() - ref_pre).pow(2).mean().sqrt().item()
    assert e3 <= e0 * 1.05 + 1e-9, (e3, e0)


@pytest.mark.parametrize('skinny', [3, 4])
def test_linear_grouped_three_problems_one_skinny_launch(skinny):
    g = torch.Generator().manual_seed(4)
    M, K = 100, 512
    x = _rand(g, M, K).to(dev())
    xp = _planes(x)
    ws = [_rand(g, n, K, scale=K ** -0.5).to(dev()) for n in (512, 512, 480)]
    bs = [_rand(g, n).to(dev()) for n in (512, 512, 480)]
    ops.set_h3_mode(skinny)
    n0 = _launches()
    outs = [torch.empty(M, w.shape[0], device=dev()) for w in ws]
    ops.linear_fwd([ops.linear_problem([(x, w, xp)], o, b) for w, o, b in zip(ws, outs, bs)])
    torch.cuda.synchronize()
    assert _launches() == n0 + 1
    for w, b, o in zip(ws, bs, outs):
        ref = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
        np.testing.assert_allclose(o.cpu().numpy(), ref.float().numpy(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('M,H,with_pre,with_tab', [(5, 64, True, False), (128, 512, True, True), (200, 512, False, False),
                                                   (33, 32, False, False)])
@pytest.mark.parametrize('skinny', [3, 4])
def test_lstm_skinny(M, H, with_pre, with_tab, skinny):
    g = torch.Generator().manual_seed(M + H)
    ks = (max(32, H // 32 * 32), 64, 32)
    xs = [_rand(g, M, k) for k in ks]
    ws = [_rand(g, 4 * H, k, scale=(3 * k) ** -0.5) for k in ks]
    b_ih, b_hh, c0 = _rand(g, 4 * H), _rand(g, 4 * H), _rand(g, M, H)
    z = sum(x.double() @ w.double().t() for x, w in zip(xs, ws)) + b_ih.double() + b_hh.double()
    kw = {}
    if with_pre:
        pre = _rand(g, M, 4 * H, scale=0.3)
        z = z + pre.double()
        kw['pre'] = pre.to(dev())
    if with_tab:
        tab = _rand(g, 50, 4 * H, scale=0.3)
        ids = torch.randint(0, 50, (M,), generator=g)
        z = z + tab.double()[ids]
        kw['tab'], kw['tab_ids'] = tab.to(dev()), ids.to(dev())
    i, f, gg, o = z.split(H, dim=1)
    c_ref = torch.sigmoid(f) * c0.double() + torch.sigmoid(i) * torch.tanh(gg)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    dx = [x.to(dev()) for x in xs]
    dsegs = [(dx[0], ws[0].to(dev()), _planes(dx[0])), (dx[1], ws[1].to(dev())), (dx[2], ws[2].to(dev()), _planes(dx[2]))]
    ops.set_h3_mode(skinny)
    n0 = _launches()
    h, c = torch.empty(M, H, device=dev()), torch.empty(M, H, device=dev())
    gates = torch.empty(M, 4 * H, device=dev())
    hp = torch.empty(2, M, H, dtype=torch.float16, device=dev()) if H % 32 == 0 else None
    ops.lstm_fwd(dsegs, b_ih.to(dev()), b_hh.to(dev()), c0.to(dev()), h, c, gates_out=gates, h_planes=hp, **kw)
    torch.cuda.synchronize()
    assert _launches() == n0 + 1
    np.testing.assert_allclose(h.cpu().numpy(), h_ref.float().numpy(), atol=2e-5)
    np.testing.assert_allclose(c.cpu().numpy(), c_ref.float().numpy(), atol=2e-5)
    act = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)], dim=1)
    np.testing.assert_allclose(gates.cpu().numpy(), act.float().numpy(), atol=2e-5)
    if hp is not None:
        assert torch.equal(hp.cpu(), _planes(h).cpu())            # the epilogue's planes == re-splitting h


@pytest.mark.parametrize('h3v', [1, 0])
@pytest.mark.parametrize('M,V,K,logits', [(5, 10000, 512, True), (128, 10000, 512, False), (70, 9487, 512, True),
                                          (33, 130, 64, True), (256, 10000, 512, False), (200, 4100, 96, True),
                                          (1, 64, 32, True), (97, 10000, 256, False)])
def test_vocab_skinny(M, V, K, logits, h3v):
    """h3v = 1: gemm_h3v_kernel (workgroup-shared k-block stages), 0: the per-wave-ring form (isc_set_h3v)."""
    prev = ops.set_h3v(h3v)
    try:
        _vocab_skinny(M, V, K, logits)
    finally:
        ops.set_h3v(prev)


def _vocab_skinny(M, V, K, logits):
    g = torch.Generator().manual_seed(V + M)
    h, W, bias = _rand(g, M, K), _rand(g, V, K, scale=4 * K ** -0.5), _rand(g, V)
    logits_ref = h.double() @ W.double().t() + bias.double()
    lse_ref = torch.logsumexp(logits_ref, 1)
    nt = (V + 127) // 128
    dh = h.to(dev())
    ops.set_h3_mode(3)
    n0 = _launches()
    pm, ps = torch.empty(M, nt, device=dev()), torch.empty(M, nt, device=dev())
    pi = torch.empty(M, nt, device=dev(), dtype=torch.int32)
    lg = torch.empty(M, V, device=dev()) if logits else None
    ops.vocab_fwd(dh, W.to(dev()), bias.to(dev()), pm, ps, pi, lg, h_planes=_planes(dh) if M % 2 else None)
    torch.cuda.synchronize()
    assert _launches() == n0 + 1
    mx = pm.max(1).values
    lse = mx + torch.log((ps * torch.exp(pm - mx[:, None])).sum(1))
    np.testing.assert_allclose(lse.cpu().numpy(), lse_ref.float().numpy(), atol=3e-5, rtol=1e-5)
    arg = pi.gather(1, pm.argmax(1)[:, None]).squeeze(1).long().cpu()
    ref_arg = logits_ref.argmax(1)
    for r in (arg != ref_arg).nonzero().flatten().tolist():
        top2 = logits_ref[r].topk(2).values
        assert (top2[0] - top2[1]).item() < 1e-5, (r, top2)
    if logits:
        np.testing.assert_allclose(lg.cpu().numpy(), logits_ref.float().numpy(), atol=3e-5, rtol=1e-5)
        assert torch.equal(arg, lg.argmax(1).cpu())


def test_auto_mode_takes_the_skinny_kernel_only_inside_a_weights_scope():
    g = torch.Generator().manual_seed(2)
    M, N, K = 128, 2048, 1024
    x, w = _rand(g, M, K).to(dev()), _rand(g, N, K, scale=K ** -0.5).to(dev())
    ref = x.double().cpu() @ w.double().cpu().t()

    def run():
        out = torch.empty(M, N, device=dev())
        ops.linear_fwd([ops.linear_problem([(x, w)], out)])
        torch.cuda.synchronize()
        return out.cpu()
    ops.set_h3_mode(1)
    n0 = _launches()
    plain = run()                                          # no scope: the fp32 split-K route, as before
    assert _launches() == n0
    with ops.h3_weights_scope(dev()):
        a, b = run(), run()                                # second call re-uses the cached weight planes
    assert _launches() == n0 + 2
    assert torch.equal(a, b) and not torch.equal(a, plain)
    np.testing.assert_allclose(a.numpy(), ref.float().numpy(), atol=2e-5, rtol=1e-5)
    ops.set_h3_mode(0)
    with ops.h3_weights_scope(dev()):
        assert torch.equal(run(), plain)                   # mode 0: exact-fp32 tiles only
    assert _launches() == n0 + 2


@pytest.mark.parametrize('M,N,K1,K2', [(128, 512, 2048, 0), (80, 1024, 512, 512), (5, 96, 64, 0), (300, 512, 2048, 0),
                                       (2560, 512, 9984, 0), (700, 96, 4096, 512)])      # long K: split over workgroups too
@pytest.mark.parametrize('skinny', [3, 4])
def test_backward_nn_contraction_on_the_skinny_kernel(M, N, K1, K2, skinny):
    """isc_gemm_bwd, NN layout (dX = dY W, W as stored [K, N]): inside a weights scope the few-row launches run on the
    skinny kernel over planes of W^T built by the transposing split; accumulate and K-segments included."""
    g = torch.Generator().manual_seed(M + N + K1)
    dy1, w1 = _rand(g, M, K1), _rand(g, K1, N + 64, scale=K1 ** -0.5)     # W is a column slice of a wider matrix
    prior = _rand(g, M, N)
    ref = prior.double() + dy1.double() @ w1.double()[:, 32:32 + N]
    segs = [(dy1.to(dev()), w1.to(dev())[:, 32:32 + N])]
    if K2:
        dy2, w2 = _rand(g, M, K2), _rand(g, K2, N, scale=K2 ** -0.5)
        ref = ref + dy2.double() @ w2.double()
        segs.append((dy2.to(dev()), w2.to(dev())))
    outs = {}
    for mode in (skinny, 0):
        ops.set_h3_mode(mode)
        n0 = _launches()
        out = prior.clone().to(dev())
        with ops.h3_weights_scope(dev()):
            ops.gemm_bwd([ops.gemm_problem(segs, out, ops.NN, accumulate=True)], ops.NN)
            ops.gemm_bwd([ops.gemm_problem(segs, out, ops.NN, accumulate=False)], ops.NN)      # planes re-used
            again = out.clone()
            out.copy_(prior)
            ops.gemm_bwd([ops.gemm_problem(segs, out, ops.NN, accumulate=True)], ops.NN)
        torch.cuda.synchronize()
        assert _launches() - n0 == (3 if mode == skinny else 0)
        np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), atol=3e-5, rtol=1e-5)
        np.testing.assert_allclose(again.cpu().numpy(), (ref - prior.double()).float().numpy(), atol=3e-5, rtol=1e-5)
        outs[mode] = (out.double().cpu() - ref).pow(2).mean().sqrt().item()
    assert outs[skinny] <= outs[0] * 1.05 + 1e-9, outs


@pytest.mark.parametrize('K1,K2,M,N', [(2560, 0, 2048, 1536), (640, 0, 512, 512), (2560, 128, 2048, 512), (12288, 0, 2048, 2048),
                                       (1600, 0, 10000, 512), (10240, 10240, 2048, 512), (1600, 80, 2048, 512)])
def test_backward_tn_contraction_on_split_f16(K1, K2, M, N):
    """isc_gemm_bwd, TN layout (dW = dY^T X: both operands [K rows, .]): planes of both TRANSPOSES are built into the
    workspace and the NT split-f16 kernels contract them - large / 64-row / skinny tiles by size, K chunks that
    accumulate when the planes exceed the workspace, two K-segments, an output with 10000 rows, and two K-segments whose
    planes exceed the workspace (the [fc | label] block of the att-LSTM's dW at B = 512: one pass per segment), and a large
    segment beside one of 80 rows (K % 32 != 0: that one accumulates on the fp32 tiles in a launch of its own)."""
    g = torch.Generator().manual_seed(K1 + M)
    a1, w1 = _rand(g, K1, M), _rand(g, K1, N, scale=K1 ** -0.5)
    prior = _rand(g, M, N)
    ref = prior.double() + a1.double().t() @ w1.double()
    segs = [(a1.to(dev()), w1.to(dev()))]
    if K2:
        a2, w2 = _rand(g, K2, M), _rand(g, K2, N, scale=K2 ** -0.5)
        ref = ref + a2.double().t() @ w2.double()
        segs.append((a2.to(dev()), w2.to(dev())))
    errs = {}
    for mode in (1, 0):
        ops.set_h3_mode(mode)
        n0 = ops._lib.load().isc_h3_launches()
        out = prior.clone().to(dev())
        ops.gemm_bwd([ops.gemm_problem(segs, out, ops.TN, accumulate=True)], ops.TN)
        torch.cuda.synchronize()
        took = ops._lib.load().isc_h3_launches() - n0
        assert took == ((2 if K1 in (12288, 10240) else 1) if mode == 1 else 0), took
        np.testing.assert_allclose(out.cpu().numpy(), ref.float().numpy(), atol=5e-5, rtol=1e-5)
        errs[mode] = (out.double().cpu() - ref).pow(2).mean().sqrt().item()
    assert errs[1] <= errs[0] * 1.05 + 1e-9, errs
