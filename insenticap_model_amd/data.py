"""Batch-tuple producers in front of the hot path.

* `create_collate_fn(name, ...)` - the four collate layouts the decoder path consumes, with the
  semantics of /root/reference/dataloader.py:8-109 (`caption`, `scs`, `rl_fact`, `rl_senti`):
  5-captions-per-image expansion, stable sort by caption length (descending), truncation to
  `max_seq_len`, <PAD> filling, `lengths - 1`.  Restated from the reference text: its module cannot be
  imported here (it needs h5py), so these are pinned by hand-checked expectations
  (tests/test_data_checkpoint.py), not by reference-generated goldens.
* `DevicePrefetcher` - pinned host staging + asynchronous H2D copies on a side HIP stream, one batch
  ahead of the consumer: at >10k captions/s the 303 KB of fp32 region features per caption
  (~3 GB/s and more) must overlap with decoding instead of serialising in front of it.
"""
import random

import numpy as np
import torch


def _pad_rows(seqs, width, pad_index, limit=None):
    out = np.full((len(seqs), width), pad_index, dtype=np.int64)
    for i, s in enumerate(seqs):
        end = min(len(s), width if limit is None else limit)
        out[i, :end] = s[:end]
    return torch.from_numpy(out)


def _caps(caps, max_seq_len, pad_index):
    lengths = [min(len(c), max_seq_len) for c in caps]
    tensor = _pad_rows(caps, lengths[0], pad_index)        # rows are sorted: lengths[0] is the maximum
    return tensor, [l - 1 for l in lengths]


def _feats(xs):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(xs, dtype=np.float32)))


def create_collate_fn(name, pad_index=0, max_seq_len=17, num_concepts=5, num_sentiments=10):
    def caption(dataset):
        rows = [(fn, fc, att, cap, cpts) for fn, fc, att, caps_idx, cpts in dataset for cap in caps_idx]
        rows.sort(key=lambda p: len(p[3]), reverse=True)          # stable, like the reference
        fns, fcs, atts, caps, cpts = zip(*rows)
        return fns, _feats(fcs), _feats(atts), _caps(caps, max_seq_len, pad_index), \
            _pad_rows(cpts, num_concepts, pad_index)

    def scs(dataset):
        rows = sorted(dataset, key=lambda p: len(p[0]), reverse=True)
        caps, cpts, sentis, senti_ids = zip(*rows)
        return _caps(caps, max_seq_len, pad_index), _pad_rows(cpts, num_concepts, pad_index), \
            _pad_rows(sentis, num_sentiments, pad_index), torch.from_numpy(np.asarray(senti_ids, dtype=np.int64))

    def rl_fact(dataset):
        ground_truth, rows = {}, []
        for fn, caps_idx, fc, att, cpts, sentis in dataset:
            ground_truth[fn] = [c[:max_seq_len] for c in caps_idx]
            rows.append((fn, random.sample(caps_idx, 1)[0], fc, att, cpts, sentis))
        rows.sort(key=lambda p: len(p[1]), reverse=True)
        fns, caps, fcs, atts, cpts, sentis = zip(*rows)
        return fns, _feats(fcs), _feats(atts), _caps(caps, max_seq_len, pad_index), \
            _pad_rows(cpts, num_concepts, pad_index), _pad_rows(sentis, num_sentiments, pad_index), ground_truth

    def rl_senti(dataset):
        fns, fcs, atts, cpts, sentis, labels = zip(*dataset)
        return fns, _feats(fcs), _feats(atts), _pad_rows(cpts, num_concepts, pad_index), \
            _pad_rows(sentis, num_sentiments, pad_index), torch.from_numpy(np.asarray(labels, dtype=np.int64))

    table = {'caption': caption, 'scs': scs, 'rl_fact': rl_fact, 'rl_senti': rl_senti}
    if name not in table:
        raise KeyError('collate %r is outside the decoder path (have: %s)' % (name, sorted(table)))
    return table[name]


def _map_tensors(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, tuple):
        return tuple(_map_tensors(o, fn) for o in obj)
    if isinstance(obj, list) and obj and torch.is_tensor(obj[0]):
        return [_map_tensors(o, fn) for o in obj]
    return obj          # file names, length lists, ground-truth dicts stay on the host


class DevicePrefetcher:
    """Wraps an iterable of collated batches: every tensor is staged in pinned memory and copied to
    `device` on a dedicated stream while the previous batch is being consumed."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        with torch.cuda.stream(self.stream):
            return _map_tensors(batch, lambda t: (t if t.is_pinned() else t.pin_memory()).to(self.device,
                                                                                             non_blocking=True))

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)      # batch `nxt` has landed
            cur = nxt
            _map_tensors(cur, lambda t: t.record_stream(torch.cuda.current_stream(self.device)) or t)
            try:
                nxt = self._stage(next(it))                                      # overlaps with the consumer
            except StopIteration:
                nxt = None
            yield cur
