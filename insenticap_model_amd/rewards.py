"""Rewards and the RL criterion of the reference (self_critical/utils.py), host side of the RL step.

* `RewardCriterion` (utils.py:169-177) - masked REINFORCE loss on [B,T] tensors.
* `get_ciderd_scorer` / `get_self_critical_reward` (utils.py:38-83) - same signatures and return
  values as the reference, computed by the native, multi-threaded CIDEr-D library
  (include/insenticap_cider.h) instead of the pure-Python scorer (~1,240 hypotheses/s/core there,
  the first wall of the RL step once decoding is fast: SURVEY 8(a-19)).
* `get_cls_reward` (utils.py:120-151) - sentence-sentiment-classifier reward.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
CIDER_LIB_PATH = os.path.join(_HERE, 'lib', 'libinsenticap_cider.so')
_cider = None


class RewardCriterion(nn.Module):
    """Masked REINFORCE loss: -sum(logp * mask * reward) / sum(mask) over [B,T] tensors (self_critical/utils.py:
    169-177).  `seq_logprobs` comes from Captioner.forward_rl (differentiable when sampling in train mode).  One HIP
    launch forward (isc_reward_loss_fwd: fixed-order sums, bit-repeatable) and one backward (isc_reward_loss_bwd) whose
    [B,T] result reaches the roll-out's BPTT as (drawn token, weight) pairs - no [B,T,V] gradient tensor exists."""

    def forward(self, seq_logprobs, seq_masks, reward):
        from .autograd import reward_criterion
        return reward_criterion(seq_logprobs, seq_masks, reward)


def _load_cider():
    global _cider
    if _cider is not None:
        return _cider
    if not os.path.exists(CIDER_LIB_PATH):
        raise RuntimeError('libinsenticap_cider.so not found at %s - run `python -m insenticap_model_amd._build`'
                           % CIDER_LIB_PATH)
    lib = C.CDLL(CIDER_LIB_PATH)
    i64p = C.POINTER(C.c_int64)
    lib.isc_cider_create.restype = C.c_void_p
    lib.isc_cider_create.argtypes = [i64p, i64p, i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_double]
    lib.isc_cider_destroy.restype = None
    lib.isc_cider_destroy.argtypes = [C.c_void_p]
    lib.isc_cider_num_images.restype = C.c_int64
    lib.isc_cider_num_images.argtypes = [C.c_void_p]
    lib.isc_cider_num_ngrams.restype = C.c_int64
    lib.isc_cider_num_ngrams.argtypes = [C.c_void_p]
    lib.isc_cider_score.restype = C.c_int
    lib.isc_cider_score.argtypes = [C.c_void_p, i64p, C.c_int64, C.c_int64, C.c_int64, i64p, i64p, i64p,
                                    C.POINTER(C.c_double), C.c_int]
    _cider = lib
    return lib


def _flatten(caption_lists):
    """[[caps of image 0], [caps of image 1], ...] -> (tokens, cap_off, img_off) int64 arrays."""
    toks, cap_off, img_off = [], [0], [0]
    for caps in caption_lists:
        for cap in caps:
            toks.extend(int(x) for x in cap)
            cap_off.append(len(toks))
        img_off.append(len(cap_off) - 1)
    return (np.asarray(toks, dtype=np.int64), np.asarray(cap_off, dtype=np.int64),
            np.asarray(img_off, dtype=np.int64))


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


class CiderD:
    """Native CIDEr-D scorer; plays the role of the reference's `CiderD(refs=...)` object
    (ciderD.py:16-48) for `get_self_critical_reward`."""

    def __init__(self, ref_caption_lists, sos_token, eos_token, n=4, sigma=6.0, n_threads=None):
        lib = _load_cider()
        toks, cap_off, img_off = _flatten(ref_caption_lists)
        self._lib = lib
        self.sos, self.eos = int(sos_token), int(eos_token)
        self._h = lib.isc_cider_create(_p(toks), _p(cap_off), _p(img_off), len(img_off) - 1, self.sos, self.eos,
                                       n, sigma)
        if not self._h:
            raise RuntimeError('isc_cider_create failed')
        # up to 16 scoring threads, of this process's share of the host's cores (one process per GPU: LOCAL_WORLD_SIZE ranks of
        # a launcher share them - eight ranks x 16 threads on a 64-core host would only queue behind each other)
        cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        try:
            local_ranks = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))
        except ValueError:
            local_ranks = 1
        self.n_threads = n_threads or max(1, min(16, cores // local_ranks))
        self._gt_cache = {}

    def __del__(self):
        if getattr(self, '_h', None):
            self._lib.isc_cider_destroy(self._h)
            self._h = None

    @property
    def num_images(self):
        return self._lib.isc_cider_num_images(self._h)

    @property
    def num_ngrams(self):
        return self._lib.isc_cider_num_ngrams(self._h)

    def flatten_refs(self, ref_caption_lists, keys=None):
        """Flattened (tokens, cap_off, img_off) of a batch of reference lists; with `keys` (image ids)
        the per-image conversion is cached across calls (the ground truth of an image never changes)."""
        if keys is None:
            return _flatten(ref_caption_lists)
        toks, lens, counts = [], [], []
        for k, caps in zip(keys, ref_caption_lists):
            c = self._gt_cache.get(k)
            if c is None:
                c = (np.asarray([int(x) for cap in caps for x in cap], dtype=np.int64),
                     np.asarray([len(cap) for cap in caps], dtype=np.int64))
                self._gt_cache[k] = c
            toks.append(c[0])
            lens.append(c[1])
            counts.append(len(c[1]))
        cap_off = np.concatenate([[0], np.cumsum(np.concatenate(lens))]).astype(np.int64)
        img_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        return np.concatenate(toks), cap_off, img_off

    def score_arrays(self, hyps, ref_caption_lists, flat=None):
        """hyps: int64 [N,T] raw roll-out rows; ref_caption_lists[i]: list of id lists (or `flat` = a
        flatten_refs() result). -> float64 [N]."""
        hyps = np.ascontiguousarray(hyps, dtype=np.int64)
        toks, cap_off, img_off = flat if flat is not None else _flatten(ref_caption_lists)
        out = np.empty(hyps.shape[0], dtype=np.float64)
        rc = self._lib.isc_cider_score(self._h, _p(hyps), hyps.shape[0], hyps.shape[1], hyps.shape[1], _p(toks),
                                       _p(cap_off), _p(img_off), out.ctypes.data_as(C.POINTER(C.c_double)),
                                       self.n_threads)
        if rc != 0:
            raise RuntimeError('isc_cider_score failed (%d)' % rc)
        return out


def get_ciderd_scorer(split_captions, sos_token, eos_token):
    """utils.py:38-53: `split_captions` = {split: {fn: [[ids], ...]}}; document frequencies over all images."""
    captions = {}
    for caps in split_captions.values():
        captions.update(caps)
    return CiderD(list(captions.values()), sos_token, eos_token)


def self_critical_scores(captions, fns, ground_truth, scorer):
    """CIDEr-D of one token matrix [B,T] (host ndarray) against the images' references -> float64 [B]: one half of
    get_self_critical_reward, for callers that score the sampled captions while the device still decodes the greedy ones."""
    if not isinstance(scorer, CiderD):
        raise Exception('do not support this scorer: %s' % type(scorer))
    assert captions.shape[0] == len(fns)
    flat = scorer.flatten_refs([ground_truth[fn] for fn in fns], keys=list(fns))
    return scorer.score_arrays(captions, None, flat)


def get_self_critical_reward(sample_captions, greedy_captions, fns, ground_truth, sos_token, eos_token, scorer):
    """utils.py:56-83: CIDEr-D(sample) - CIDEr-D(greedy), repeated over T -> float64 ndarray [B,T]."""
    batch_size = len(fns)
    if torch.is_tensor(sample_captions):
        sample_captions = sample_captions.cpu().numpy()
    if torch.is_tensor(greedy_captions):
        greedy_captions = greedy_captions.cpu().numpy()
    assert sample_captions.shape[0] == greedy_captions.shape[0] == batch_size
    if not isinstance(scorer, CiderD):
        raise Exception('do not support this scorer: %s' % type(scorer))
    flat = scorer.flatten_refs([ground_truth[fn] for fn in fns], keys=list(fns))
    scores = scorer.score_arrays(sample_captions, None, flat) - scorer.score_arrays(greedy_captions, None, flat)
    return np.repeat(scores[:, np.newaxis], sample_captions.shape[1], 1)


def get_cls_reward(sample_captions, sample_masks, greedy_captions, greedy_masks, senti_labels, sent_senti_cls,
                   sample_lens=None, on_device=False):
    """utils.py:120-151: 1[classifier(sample) == label] x per-token squeeze-excite weights, zero-padded
    to T. `sent_senti_cls` is the frozen helper net (helper_nets.SentenceSentimentClassifier).
    `sample_lens` (host ints) skips the device->host read of the mask sums - or a device tensor [B]: the classifier then
    runs over all T columns and needs no host value at all (the launches can be enqueued before the roll-out has
    finished); `on_device=True` returns the [B,T] float tensor without a host round trip (the trainer adds it to the
    CIDEr reward on the device)."""
    training = sent_senti_cls.training
    if sample_lens is None:
        sample_lens = list(sample_masks.sum(dim=-1).type(torch.int).cpu().numpy())
    sent_senti_cls.eval()
    with torch.no_grad():
        sample_preds, sample_att_weights = sent_senti_cls(sample_captions, sample_lens)
        sample_preds = sample_preds.softmax(dim=-1).argmax(dim=-1)
        sample_preds = (sample_preds == senti_labels).type_as(sample_att_weights).unsqueeze(1)
        sample_scores = (sample_preds * sample_att_weights).detach()
    sent_senti_cls.train(training)
    max_len = sample_captions.shape[1]
    if on_device:
        return torch.nn.functional.pad(sample_scores, (0, max_len - sample_scores.shape[1]))
    sample_scores = sample_scores.cpu().numpy()
    return np.pad(sample_scores, ((0, 0), (0, max_len - sample_scores.shape[1])))
