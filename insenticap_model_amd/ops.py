"""Thin torch-tensor -> C-ABI adapters. PyTorch is plumbing here (device memory + streams);
all arithmetic happens in libinsenticap_hip.so."""
import ctypes as C

import torch

from . import _lib
from ._lib import LinearProblem, LstmProblem, RolloutStep, ScanProblem, check


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the stream the kernels run on
    (torch's current stream). bench.py arms it for one decode step per timed roll-out; when
    `armed` is False the wrappers below add no work."""

    def __init__(self):
        self.armed = False
        self.arm_step = None   # decode step whose kernels get timed (-1 = prologue, None = off)
        self.phase = 'step'    # 'prologue' kernels run once per roll-out, 'step' kernels T times
        self.records = []   # (name, start_event, end_event, flops, bytes)

    def begin(self):
        if not self.armed:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, e0, name, flops=0.0, nbytes=0.0):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((name, e0, e1, flops, nbytes, self.phase))

    def summary(self):
        """name -> dict(n, avg_ms, flops, bytes); call after a device synchronize."""
        out = {}
        for name, e0, e1, fl, nb, ph in self.records:
            d = out.setdefault(name, dict(n=0, total_ms=0.0, flops=fl, bytes=nb, phase=ph))
            d['n'] += 1
            d['total_ms'] += e0.elapsed_time(e1)
        for d in out.values():
            d['avg_ms'] = d['total_ms'] / d['n']
        return out


TIMER = KernelTimer()


def _seg_k(segs):
    return sum(a.shape[1] for a, _ in segs)


def ptr(t):
    return None if t is None else t.data_ptr()


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.HipLibraryError(
                'insenticap_model_amd runs on a ROCm device only (got a CPU tensor); there is no CPU path')


def _fill_segs(dst, segs):
    if not 1 <= len(segs) <= _lib.ISC_MAX_SEG:
        raise ValueError('1..4 K-segments supported')
    for i, (A, W) in enumerate(segs):
        assert A.dim() == 2 and W.dim() == 2 and A.stride(1) == 1 and W.stride(1) == 1
        assert A.shape[1] == W.shape[1], (A.shape, W.shape)
        assert A.dtype == torch.float32 and W.dtype == torch.float32
        s = dst[i]
        s.A, s.W = A.data_ptr(), W.data_ptr()
        s.lda, s.ldw, s.K = A.stride(0), W.stride(0), A.shape[1]


def linear_problem(segs, out, bias0=None, bias1=None, bias2=None, relu=False, keep_mask=None, mask_scale=1.0,
                   out_pre=None):
    """out[M,N] = act(sum_s A_s W_s^T + bias0 + bias1) [* keep_mask * mask_scale]."""
    p = LinearProblem()
    _fill_segs(p.seg, segs)
    p.nseg = len(segs)
    p.M, p.N = out.shape
    assert segs[0][0].shape[0] == p.M and segs[0][1].shape[0] == p.N
    assert out.stride(1) == 1
    p.relu = int(relu)
    p.bias0, p.bias1, p.bias2 = ptr(bias0), ptr(bias1), ptr(bias2)
    p.keep_mask = ptr(keep_mask)
    p.mask_scale = mask_scale
    p.ldc = out.stride(0)
    p.C = out.data_ptr()
    p.C_pre = ptr(out_pre)
    if out_pre is not None:
        assert out_pre.stride(0) == out.stride(0)
    return p


def linear_fwd(problems):
    lib = _lib.load()
    arr = (LinearProblem * len(problems))(*problems)
    e0 = TIMER.begin()
    check(lib.isc_linear_fwd(arr, len(problems), stream()), 'isc_linear_fwd')
    if e0 is not None:
        fl = sum(2.0 * q.M * q.N * sum(q.seg[i].K for i in range(q.nseg)) for q in problems)
        TIMER.end(e0, 'linear[' + '+'.join('%dx%dx%d' % (q.M, q.N, sum(q.seg[i].K for i in range(q.nseg)))
                                          for q in problems) + ']', fl)


def lstm_fwd(segs, b_ih, b_hh, c_prev, h_out, c_out, gates_out=None, h_keep_mask=None,
             mask_scale=1.0, hdrop_out=None):
    lib = _lib.load()
    p = LstmProblem()
    _fill_segs(p.seg, segs)
    p.nseg = len(segs)
    p.M, p.H = c_prev.shape
    assert c_prev.is_contiguous() and h_out.is_contiguous() and c_out.is_contiguous()
    p.b_ih, p.b_hh = b_ih.data_ptr(), b_hh.data_ptr()
    p.c_prev, p.h_out, p.c_out = c_prev.data_ptr(), h_out.data_ptr(), c_out.data_ptr()
    p.gates_out = ptr(gates_out)
    p.h_keep_mask = ptr(h_keep_mask)
    p.mask_scale = mask_scale
    p.hdrop_out = ptr(hdrop_out)
    e0 = TIMER.begin()
    check(lib.isc_lstm_fwd(C.byref(p), stream()), 'isc_lstm_fwd')
    if e0 is not None:
        k = _seg_k(segs)
        TIMER.end(e0, 'lstm[%dx%dx%d]' % (p.M, 4 * p.H, k), 2.0 * p.M * 4 * p.H * k)


def vocab_fwd(h, W, bias, part_max, part_sum, part_idx, logits=None):
    lib = _lib.load()
    M, K = h.shape
    V = W.shape[0]
    assert h.stride(1) == 1 and W.stride(1) == 1
    ld = logits.stride(0) if logits is not None else 0
    e0 = TIMER.begin()
    check(lib.isc_vocab_fwd(h.data_ptr(), h.stride(0), W.data_ptr(), W.stride(0), bias.data_ptr(), M, V, K,
                            ptr(logits), ld, part_max.data_ptr(), part_sum.data_ptr(),
                            part_idx.data_ptr(), stream()), 'isc_vocab_fwd')
    TIMER.end(e0, 'vocab[%dx%dx%d]' % (M, V, K), 2.0 * M * V * K)


def logsoftmax_apply(logits, part_max, part_sum, lse_out=None):
    lib = _lib.load()
    M, V = logits.shape
    assert logits.stride(1) == 1
    check(lib.isc_logsoftmax_apply(logits.data_ptr(), logits.stride(0), M, V, part_max.data_ptr(),
                                   part_sum.data_ptr(), ptr(lse_out), stream()), 'isc_logsoftmax_apply')


def scan_problem(P, V, q, w, w_bias, out, alpha_out=None, q2=None):
    """P [B,R,A], V [B,R,D] contiguous; q/q2 [B,A]; w [A] (or [1,A]); out [B,D];
    alpha_out: [B,R] view with unit inner stride (row stride arbitrary)."""
    s = ScanProblem()
    assert P.is_contiguous() and V.is_contiguous() and q.is_contiguous() and out.is_contiguous()
    s.P, s.V, s.q, s.q2, s.w = P.data_ptr(), V.data_ptr(), q.data_ptr(), ptr(q2), w.data_ptr()
    s.w_bias = ptr(w_bias)
    s.R, s.A, s.D = P.shape[1], P.shape[2], V.shape[2]
    s.out = out.data_ptr()
    if alpha_out is not None:
        assert alpha_out.stride(1) == 1
        s.alpha_out, s.alpha_ld = alpha_out.data_ptr(), alpha_out.stride(0)
    return s


def attn_scan_fwd(problems, B):
    lib = _lib.load()
    arr = (ScanProblem * len(problems))(*problems)
    e0 = TIMER.begin()
    check(lib.isc_attn_scan_fwd(arr, len(problems), B, stream()), 'isc_attn_scan_fwd')
    if e0 is not None:
        # algorithmic bytes: P and V streamed once per row (B*2*R*E*4 for the content scan)
        nb = sum(4.0 * B * q.R * (q.A + q.D) for q in problems)
        TIMER.end(e0, 'attn_scan[' + '+'.join('%dx%dx%d' % (B, q.R, q.A) for q in problems) + ']', 0.0, nb)


def gate_mix_fwd(z, w, w_bias, v, s, out, beta_out=None):
    lib = _lib.load()
    B, A = z.shape
    D = v.shape[1]
    assert z.is_contiguous() and v.is_contiguous() and s.is_contiguous() and out.is_contiguous()
    bl = beta_out.stride(0) if beta_out is not None else 0
    e0 = TIMER.begin()
    check(lib.isc_gate_mix_fwd(z.data_ptr(), w.data_ptr(), ptr(w_bias), v.data_ptr(), s.data_ptr(), B, A, D,
                               out.data_ptr(), ptr(beta_out), bl, stream()), 'isc_gate_mix_fwd')
    TIMER.end(e0, 'gate_mix[%dx%d]' % (B, A), 0.0, 4.0 * B * (A + 3 * D))


def embed_relu_fwd(emb, ids, out, add=None):
    lib = _lib.load()
    V, W = emb.shape
    assert ids.dtype == torch.int64 and ids.dim() == 1 and out.is_contiguous()
    check(lib.isc_embed_relu_fwd(emb.data_ptr(), V, W, ids.data_ptr(), ids.stride(0), ptr(add),
                                 ids.shape[0], out.data_ptr(), stream()), 'isc_embed_relu_fwd')


def embed_relu_mean_fwd(emb, ids, out):
    lib = _lib.load()
    V, W = emb.shape
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    B, Cn = ids.shape
    check(lib.isc_embed_relu_mean_fwd(emb.data_ptr(), V, W, ids.data_ptr(), Cn, B, out.data_ptr(),
                                      stream()), 'isc_embed_relu_mean_fwd')


def embed_senti_words_fwd(emb, ids, pad_id, out, keep_mask=None, mask_scale=1.0):
    lib = _lib.load()
    V, W = emb.shape
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    B, n = ids.shape
    check(lib.isc_embed_senti_words_fwd(emb.data_ptr(), V, W, ids.data_ptr(), n, pad_id, B,
                                        ptr(keep_mask), mask_scale, out.data_ptr(), stream()),
          'isc_embed_senti_words_fwd')


def rollout_finalize(step):
    lib = _lib.load()
    e0 = TIMER.begin()
    check(lib.isc_rollout_finalize(C.byref(step), stream()), 'isc_rollout_finalize')
    TIMER.end(e0, 'rollout_finalize[%d]' % step.B, 0.0, 4.0 * step.B * (3 * step.n_tile + 2 * step.W))


def beam_topk(logits, part_max, part_sum, last_word, beam, pad_id, sos_id, unk_id, mask_special,
              decoding_constraint, top_val, top_idx):
    lib = _lib.load()
    rows, V = logits.shape
    n_tile = part_max.shape[1]
    check(lib.isc_beam_topk(logits.data_ptr(), logits.stride(0), part_max.data_ptr(), part_sum.data_ptr(),
                            n_tile, rows, V, beam, last_word.data_ptr(), pad_id, sos_id, unk_id,
                            int(mask_special), int(decoding_constraint), top_val.data_ptr(),
                            top_idx.data_ptr(), stream()), 'isc_beam_topk')


def xe_loss_fwd(logp, target, lengths_i32, out2):
    lib = _lib.load()
    B, T, V = logp.shape
    assert logp.is_contiguous() and target.is_contiguous()
    check(lib.isc_xe_loss_fwd(logp.data_ptr(), target.data_ptr(), lengths_i32.data_ptr(), B, T, V,
                              out2.data_ptr(), stream()), 'isc_xe_loss_fwd')


__all__ = [n for n in dir() if n.endswith('_fwd') or n.endswith('_problem')] + [
    'RolloutStep', 'rollout_finalize', 'beam_topk', 'logsoftmax_apply', 'require_device']
