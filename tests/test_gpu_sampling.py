"""Device-side inverse-CDF sampling of the sampled roll-out (isc_rollout_finalize with `sample_u`;
reference: torch.multinomial(exp(logprobs), 1) at captioner.py:333-335).  The draw for a given uniform u is
the first vocabulary id whose cumulative softmax mass exceeds u."""
import ctypes as C

import numpy as np
import pytest
import torch

from insenticap_model_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _sample(logits, u, eos_id=2):
    dev = logits.device
    B, V = logits.shape
    W, T, nt = 32, 1, (V + 127) // 128
    pm, ps = torch.empty(B, nt, device=dev), torch.empty(B, nt, device=dev)
    pi = torch.empty(B, nt, device=dev, dtype=torch.int32)
    # the tile statistics come from the vocabulary kernel: give it h = I_B (K padded to 32) and W^T = logits, so
    # its output IS the wanted logits matrix (exactly: one non-zero product per element)
    K = ((B + 31) // 32) * 32
    h = torch.zeros(B, K, device=dev)
    h[torch.arange(B), torch.arange(B)] = 1.0
    Wm = torch.zeros(V, K, device=dev)
    Wm[:, :B] = logits.t()
    out = torch.empty(B, V, device=dev)
    ops.vocab_fwd(h, Wm, torch.zeros(V, device=dev), pm, ps, pi, out)
    st = _lib.RolloutStep()
    st.B, st.V, st.T, st.t, st.n_tile, st.W = B, V, T, 0, nt, W
    seq = torch.zeros(B, T, dtype=torch.int64, device=dev)
    lp, mk = torch.zeros(B, T, device=dev), torch.zeros(B, T, device=dev)
    unf = torch.ones(B, dtype=torch.int32, device=dev)
    alive = torch.tensor([B, 0], dtype=torch.int32, device=dev)
    emb = torch.zeros(V, W, device=dev)
    uu = u.to(dev).float().view(B, T).contiguous()
    st.part_max, st.part_sum, st.part_idx = pm.data_ptr(), ps.data_ptr(), pi.data_ptr()
    st.logits, st.ld_logits = out.data_ptr(), out.stride(0)
    st.forced, st.sample_u, st.eos_id = None, uu.data_ptr(), eos_id
    st.seq, st.seq_logprobs, st.seq_masks = seq.data_ptr(), lp.data_ptr(), mk.data_ptr()
    st.unfinished, st.alive, st.raw_tokens = unf.data_ptr(), alive.data_ptr(), None
    st.emb, st.xt_add, st.xt_next = emb.data_ptr(), None, None
    ops.rollout_finalize(st)
    torch.cuda.synchronize()
    return seq[:, 0].cpu().numpy(), lp[:, 0].cpu().numpy(), out.cpu()


@pytest.mark.parametrize('V', [1000, 10000, 130])
def test_inverse_cdf_matches_fp64(V):
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(V)
    B = 512
    logits = (torch.randn(B, V, generator=g) * 2.5).to(dev)
    u = torch.rand(B, generator=g)
    u[:4] = torch.tensor([0.0, 1e-9, 0.999999, 0.5])
    tok, lp, x = _sample(logits, u)
    p = torch.softmax(x.double(), 1)
    cdf = torch.cumsum(p, 1).numpy()
    un = u.double().numpy()
    ref = np.minimum((cdf > un[:, None]).argmax(1), V - 1)
    # fp32 partial sums may move a draw across a boundary only when u sits within rounding of it
    lo = np.where(tok > 0, cdf[np.arange(B), np.maximum(tok - 1, 0)], 0.0)
    hi = cdf[np.arange(B), tok]
    assert ((un >= lo - 2e-6) & (un <= hi + 2e-6)).all()
    assert (tok == ref).mean() > 0.995
    np.testing.assert_allclose(lp, torch.log_softmax(x.double(), 1).numpy()[np.arange(B), tok], atol=2e-5)


def test_sampling_frequencies_follow_softmax():
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(3)
    V, B = 300, 4096
    row = torch.randn(V, generator=g) * 1.5
    logits = row.repeat(B, 1).to(dev)
    counts = np.zeros(V)
    for rep in range(4):
        tok, _, _ = _sample(logits, torch.rand(B, generator=g))
        counts += np.bincount(tok, minlength=V)
    p = torch.softmax(row.double(), 0).numpy()
    n = counts.sum()
    z = (counts - n * p) / np.sqrt(n * p * (1 - p) + 1e-12)
    assert np.abs(z).max() < 5.5 and np.abs(z[p * n > 20]).mean() < 1.2


def test_scheduled_sampling_kernel():
    """isc_sched_sample: unselected rows keep the ground-truth token, selected rows get the inverse-CDF draw of
    exp(logp) for their uniform; the selection itself is u_select < ss_prob."""
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    B, V, K = 300, 1000, 64
    h, W, bias = (torch.randn(B, K, generator=g)).to(dev), (torch.randn(V, K, generator=g) * 0.4).to(dev), torch.zeros(V, device=dev)
    nt = (V + 127) // 128
    pm, ps = torch.empty(B, nt, device=dev), torch.empty(B, nt, device=dev)
    pi = torch.empty(B, nt, device=dev, dtype=torch.int32)
    logits = torch.empty(B, V, device=dev)
    ops.vocab_fwd(h, W, bias, pm, ps, pi, logits)
    ops.logsoftmax_apply(logits, pm, ps)                       # now log-probs, tile statistics unchanged
    caps = torch.randint(4, V, (B, 7), generator=g).to(dev)    # base ids = a strided column
    u_sel, u_draw = torch.rand(B, generator=g).to(dev), torch.rand(B, generator=g).to(dev)
    out = torch.empty(B, dtype=torch.int64, device=dev)
    ops.sched_sample(logits, pm, ps, pi, u_sel, u_draw, 0.4, caps[:, 3], out)
    torch.cuda.synchronize()
    sel = (u_sel < 0.4).cpu().numpy()
    got, base = out.cpu().numpy(), caps[:, 3].cpu().numpy()
    assert (got[~sel] == base[~sel]).all() and 0.25 < sel.mean() < 0.55
    cdf = torch.cumsum(logits.double().exp(), 1).cpu().numpy()
    un = u_draw.double().cpu().numpy()
    lo = np.where(got > 0, cdf[np.arange(B), np.maximum(got - 1, 0)], 0.0)
    hi = cdf[np.arange(B), got]
    assert ((un >= lo - 2e-6) & (un <= hi + 2e-6))[sel].all()
