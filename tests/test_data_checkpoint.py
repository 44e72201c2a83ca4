"""Batch-tuple producers (data.py) and reference-format checkpoints / result files (checkpoint.py)."""
import json
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from insenticap_model_amd import Captioner, checkpoint, clip_gradient, data, synth


# ----------------------------------------------------------------------------- collates
def _img(fn, caps, ncpt=3, nsent=4, F=8, R=3):
    rng = np.random.default_rng(abs(hash(fn)) % 1000)
    return fn, rng.random(F, dtype=np.float32), rng.random((R, F), dtype=np.float32), caps, \
        list(range(10, 10 + ncpt)), list(range(20, 20 + nsent))


def test_caption_collate_expands_sorts_truncates():
    f = data.create_collate_fn('caption', pad_index=0, max_seq_len=5, num_concepts=4)
    a = _img('a', [[1, 5, 6, 2], [1, 7, 8, 9, 10, 11, 2]])
    b = _img('b', [[1, 4, 2]])
    ds = [(a[0], a[1], a[2], a[3], a[4]), (b[0], b[1], b[2], b[3], b[4])]
    fns, fc, att, (caps, lengths), cpts = f(ds)
    assert fns == ('a', 'a', 'b')                          # one row per caption, longest first (stable)
    assert caps.tolist() == [[1, 7, 8, 9, 10], [1, 5, 6, 2, 0], [1, 4, 2, 0, 0]]
    assert lengths == [4, 3, 2]                            # min(len, max_seq_len) - 1
    assert caps.dtype == torch.int64 and fc.dtype == torch.float32 and fc.shape == (3, 8) and att.shape == (3, 3, 8)
    assert cpts.tolist() == [[10, 11, 12, 0]] * 3          # padded to num_concepts
    assert torch.equal(fc[0], fc[1])                       # both rows of image a share its features


def test_caption_width_pads_beyond_the_longest_caption_and_leaves_the_lengths():
    """caption_width (not in the reference): 'full' pads every batch to max_seq_len, an int m rounds the unroll length
    (width - 1) up to a multiple of m - one input geometry for the graph-served training steps whatever the longest caption
    of a batch is; rows and lengths are those of the tight collate."""
    a = _img('a', [[1, 5, 6, 2], [1, 7, 8, 2]])
    b = _img('b', [[1, 4, 2]])
    ds = [(a[0], a[1], a[2], a[3], a[4]), (b[0], b[1], b[2], b[3], b[4])]
    tight = data.create_collate_fn('caption', pad_index=0, max_seq_len=9, num_concepts=4)(ds)
    assert tight[3][0].shape == (3, 4) and tight[3][1] == [3, 3, 2]
    full = data.create_collate_fn('caption', pad_index=0, max_seq_len=9, num_concepts=4, caption_width='full')(ds)
    assert full[3][0].shape == (3, 9) and full[3][1] == tight[3][1]
    assert full[3][0][:, :4].tolist() == tight[3][0].tolist() and int(full[3][0][:, 4:].abs().sum()) == 0
    by4 = data.create_collate_fn('caption', pad_index=0, max_seq_len=9, num_concepts=4, caption_width=4)(ds)
    assert by4[3][0].shape == (3, 5) and by4[3][1] == tight[3][1]          # unroll length 3 -> 4 steps (+ <SOS>)
    capped = data.create_collate_fn('caption', pad_index=0, max_seq_len=4, num_concepts=4, caption_width=8)(ds)
    assert capped[3][0].shape == (3, 4)                                     # never beyond max_seq_len


def test_caption_collate_dedup_hands_each_image_over_once():
    """dedup=True (not in the reference): the 'caption' collate makes one row per caption, so an image's features repeat;
    as data.RowGather (distinct rows + a row index) they are a quarter of the bytes, and dense() is the reference layout."""
    a = _img('a', [[1, 5, 6, 2], [1, 7, 8, 9, 10, 11, 2], [1, 3, 2]])
    b = _img('b', [[1, 4, 2], [1, 4, 4, 4, 2]])
    ds = [(a[0], a[1], a[2], a[3], a[4]), (b[0], b[1], b[2], b[3], b[4])]
    plain = data.create_collate_fn('caption', pad_index=0, max_seq_len=9, num_concepts=4)(ds)
    dd = data.create_collate_fn('caption', pad_index=0, max_seq_len=9, num_concepts=4, dedup=True)(ds)
    assert isinstance(dd[1], data.RowGather) and dd[1].base.shape[0] == 2 and dd[2].base.shape[0] == 2
    assert dd[1].shape == tuple(plain[1].shape) and dd[2].shape == tuple(plain[2].shape)
    assert torch.equal(dd[1].dense(), plain[1]) and torch.equal(dd[2].dense(), plain[2])
    assert dd[0] == plain[0] and torch.equal(dd[3][0], plain[3][0]) and dd[3][1] == plain[3][1] and torch.equal(dd[4], plain[4])


def test_scs_and_rl_collates():
    f = data.create_collate_fn('scs', max_seq_len=6, num_concepts=2, num_sentiments=3)
    (caps, lengths), cpts, sentis, ids = f([([1, 9, 2], [5, 6, 7], [8], 1), ([1, 3, 4, 5, 2], [5], [8, 9, 9, 9], 0)])
    assert caps.tolist() == [[1, 3, 4, 5, 2], [1, 9, 2, 0, 0]] and lengths == [4, 2]
    assert cpts.tolist() == [[5, 0], [5, 6]] and sentis.tolist() == [[8, 9, 9], [8, 0, 0]] and ids.tolist() == [0, 1]

    random.seed(3)
    g = data.create_collate_fn('rl_fact', max_seq_len=4, num_concepts=3, num_sentiments=4)
    a, b = _img('a', [[1, 5, 6, 7, 8, 2], [1, 9, 2]]), _img('b', [[1, 4, 4, 2]])
    out = g([(a[0], a[3], a[1], a[2], a[4], a[5]), (b[0], b[3], b[1], b[2], b[4], b[5])])
    fns, fc, att, (caps, lengths), cpts, sentis, gt = out
    assert gt == {'a': [[1, 5, 6, 7], [1, 9, 2]], 'b': [[1, 4, 4, 2]]}          # truncated to max_seq_len
    assert set(fns) == {'a', 'b'} and caps.shape[0] == 2 and caps.shape[1] == lengths[0] + 1
    assert all(lengths[i] >= lengths[i + 1] for i in range(len(lengths) - 1))
    assert sentis.shape == (2, 4) and cpts.shape == (2, 3)

    h = data.create_collate_fn('rl_senti', num_concepts=3, num_sentiments=2)
    fns, fc, att, cpts, sentis, labels = h([(a[0], a[1], a[2], a[4], a[5], 2), (b[0], b[1], b[2], b[4], b[5], 0)])
    assert fns == ('a', 'b') and labels.tolist() == [2, 0] and sentis.tolist() == [[20, 21], [20, 21]]
    with pytest.raises(KeyError):
        data.create_collate_fn('concept')                  # outside the decoder path


def _ragged(g, key):
    flat, lens = g[key + '/flat'], g[key + '/len']
    off = np.concatenate([[0], np.cumsum(lens)])
    return [[int(x) for x in flat[off[i]:off[i + 1]]] for i in range(len(lens))]


def _collate_inputs(g):
    fns = [str(x) for x in g['in/fns']]
    caps = _ragged(g, 'in/caps')
    caps5 = [caps[5 * i:5 * i + 5] for i in range(len(fns))]
    return fns, g['in/fc'], g['in/att'], caps5, _ragged(g, 'in/cpts'), _ragged(g, 'in/sentis'), \
        [int(x) for x in g['in/labels']]


def test_collates_vs_reference_goldens(golden, tmp_path):
    """tests/golden/collate.npz holds the outputs of the reference's OWN collate functions (dataloader.py:11-109) on
    ragged inputs that exercise truncation, <PAD> filling, length ties (stable sort) and the 5-captions expansion;
    the same inputs through data.py - directly, and through the h5-free datasets + loader factories."""
    g = golden('collate')
    fns, fcs, atts, caps5, cpts, sentis, labels = _collate_inputs(g)
    n = len(fns)
    kw = dict(pad_index=0, max_seq_len=9, num_concepts=5, num_sentiments=10)

    def same(t, key):
        assert t.dtype == (torch.int64 if g[key].dtype == np.int64 else torch.float32), key
        np.testing.assert_array_equal(t.numpy(), g[key], err_msg=key)

    r = data.create_collate_fn('caption', **kw)([(fns[i], fcs[i], atts[i], caps5[i], cpts[i]) for i in range(n)])
    assert list(r[0]) == [str(x) for x in g['caption/fns']]
    same(r[1], 'caption/fc'); same(r[2], 'caption/att'); same(r[3][0], 'caption/caps'); same(r[4], 'caption/cpts')
    assert r[3][1] == g['caption/lengths'].tolist()

    for name in ('senti_corpus_with_sentis', 'scs'):
        r = data.create_collate_fn(name, **kw)([(caps5[i][0], cpts[i], sentis[i], labels[i]) for i in range(n)])
        same(r[0][0], 'scs/caps'); same(r[1], 'scs/cpts'); same(r[2], 'scs/sentis'); same(r[3], 'scs/labels')
        assert r[0][1] == g['scs/lengths'].tolist()

    random.seed(31337)                           # the reference draws the XE caption with random.sample
    r = data.create_collate_fn('rl_fact', **kw)([(fns[i], caps5[i], fcs[i], atts[i], cpts[i], sentis[i])
                                                 for i in range(n)])
    assert list(r[0]) == [str(x) for x in g['rl_fact/fns']]
    same(r[1], 'rl_fact/fc'); same(r[2], 'rl_fact/att'); same(r[3][0], 'rl_fact/caps')
    same(r[4], 'rl_fact/cpts'); same(r[5], 'rl_fact/sentis')
    assert r[3][1] == g['rl_fact/lengths'].tolist()
    gt_rows, cnt = _ragged(g, 'rl_fact/gt'), g['rl_fact/gt_count']
    off = np.concatenate([[0], np.cumsum(cnt)])
    assert r[6] == {fn: gt_rows[off[i]:off[i + 1]] for i, fn in enumerate(fns)}

    r = data.create_collate_fn('rl_senti', **kw)([(fns[i], fcs[i], atts[i], cpts[i], sentis[i], labels[i])
                                                  for i in range(n)])
    assert list(r[0]) == [str(x) for x in g['rl_senti/fns']]
    same(r[1], 'rl_senti/fc'); same(r[2], 'rl_senti/att'); same(r[3], 'rl_senti/cpts')
    same(r[4], 'rl_senti/sentis'); same(r[5], 'rl_senti/labels')

    # the h5-free datasets (memory-mapped feature stores) + loader factories reproduce the same batches
    fc_path = data.FeatureStore.write(str(tmp_path / 'fc.npy'), fns, fcs)
    att_path = data.FeatureStore.write(str(tmp_path / 'att.npy'), fns, atts)
    captions = {fn: caps5[i] for i, fn in enumerate(fns)}
    det_c, det_s = dict(zip(fns, cpts)), dict(zip(fns, sentis))
    (b,) = list(data.get_caption_dataloader(fc_path, att_path, captions, det_c, 0, 8, 5, batch_size=n, shuffle=False))
    assert list(b[0]) == [str(x) for x in g['caption/fns']]
    same(b[1], 'caption/fc'); same(b[2], 'caption/att'); same(b[3][0], 'caption/caps'); same(b[4], 'caption/cpts')
    random.seed(31337)
    (b,) = list(data.get_rl_fact_dataloader(fc_path, att_path, captions, det_c, det_s, 0, 8, 5, 10, batch_size=n,
                                            shuffle=False))
    same(b[2], 'rl_fact/att'); same(b[3][0], 'rl_fact/caps'); same(b[5], 'rl_fact/sentis')
    (b,) = list(data.get_rl_senti_dataloader(fc_path, att_path, det_c, det_s, list(zip(fns, labels)), 0, 5, 10,
                                             batch_size=n, shuffle=False))
    same(b[1], 'rl_senti/fc'); same(b[5], 'rl_senti/labels')
    (b,) = list(data.get_senti_corpus_with_sentis_dataloader(
        [(caps5[i][0], cpts[i], sentis[i], labels[i]) for i in range(n)], 0, 8, 5, 10, batch_size=n, shuffle=False))
    same(b[0][0], 'scs/caps'); same(b[3], 'scs/labels')
    # two workers, several batches: every image is delivered exactly once
    seen = []
    for b in data.get_rl_senti_dataloader(fc_path, att_path, det_c, det_s, list(zip(fns, labels)), 0, 5, 10,
                                          batch_size=3, num_workers=2, shuffle=True):
        seen.extend(b[0])
    assert sorted(seen) == sorted(fns)


class _FakeH5File:
    """h5py's surface as dataloader.py:171-178 uses it - File(path, mode='r')[fn][:] - over an .npz (h5py is not in this
    image; the h5 path's parity against real h5 files is unpinned, the store's logic is what this covers)."""
    opened = 0

    def __init__(self, path, mode='r'):
        assert mode == 'r'
        type(self).opened += 1
        self._z = np.load(path + '.npz')

    def __getitem__(self, fn):
        return self._z[fn]                    # ndarray: `[:]` works as on an h5 dataset

    def __contains__(self, fn):
        return fn in self._z.files

    def __len__(self):
        return len(self._z.files)

    def keys(self):
        return list(self._z.files)


def test_reference_h5_feature_files_are_taken_as_they_are(golden, tmp_path, monkeypatch):
    """A caller that hands the Dataset classes the `.h5` paths it handed the reference's (dataloader.py:164-178) gets the
    same batches as from the .npy stores: data.H5FeatureStore (one open per process instead of two per item), and the
    one-off conversion data.FeatureStore.from_h5.  Without h5py the call says what to do instead of failing inside numpy."""
    import sys
    import types
    g = golden('collate')
    fns, fcs, atts, caps5, cpts, sentis, labels = _collate_inputs(g)
    n = len(fns)
    fc_h5, att_h5 = str(tmp_path / 'fc_feats.h5'), str(tmp_path / 'att_feats.hdf5')
    np.savez(fc_h5 + '.npz', **{fn: fcs[i] for i, fn in enumerate(fns)})
    np.savez(att_h5 + '.npz', **{fn: atts[i] for i, fn in enumerate(fns)})
    captions = {fn: caps5[i] for i, fn in enumerate(fns)}
    det_c = dict(zip(fns, cpts))

    monkeypatch.setitem(sys.modules, 'h5py', None)               # h5py absent: a loud, actionable error
    with pytest.raises(ImportError, match='FeatureStore.from_h5'):
        data.get_caption_dataloader(fc_h5, att_h5, captions, det_c, 0, 8, 5, batch_size=n, shuffle=False)

    fake = types.ModuleType('h5py')
    fake.File = _FakeH5File
    monkeypatch.setitem(sys.modules, 'h5py', fake)
    _FakeH5File.opened = 0
    (b,) = list(data.get_caption_dataloader(fc_h5, att_h5, captions, det_c, 0, 8, 5, batch_size=n, shuffle=False))
    assert _FakeH5File.opened == 2                              # one open per file, not two per item
    assert list(b[0]) == [str(x) for x in g['caption/fns']]
    for t, key in ((b[1], 'caption/fc'), (b[2], 'caption/att'), (b[3][0], 'caption/caps'), (b[4], 'caption/cpts')):
        np.testing.assert_array_equal(t.numpy(), g[key], err_msg=key)
    # a store travels to loader workers as its path (the handle is re-opened there)
    import pickle
    st = data.H5FeatureStore(fc_h5)
    st2 = pickle.loads(pickle.dumps(st))
    assert st2._file is None and fns[0] in st2 and len(st2) == n
    np.testing.assert_array_equal(st2[fns[1]], fcs[1])
    # one-off conversion: same rows from the memory-mapped pair
    conv = data.FeatureStore.from_h5(att_h5, str(tmp_path / 'att_conv'))
    assert sorted(conv.keys()) == sorted(fns)
    for i, fn in enumerate(fns):
        np.testing.assert_array_equal(conv[fn], atts[i])
    (b2,) = list(data.get_caption_dataloader(fc_h5, conv.path, captions, det_c, 0, 8, 5, batch_size=n, shuffle=False))
    np.testing.assert_array_equal(b2[2].numpy(), g['caption/att'])


@pytest.mark.gpu
def test_device_prefetcher_delivers_identical_batches():
    dev = torch.device('cuda:0')
    f = data.create_collate_fn('caption', max_seq_len=6)
    batches = []
    for k in range(5):
        ds = [(x[0], x[1], x[2], x[3], x[4]) for x in (_img('i%d_%d' % (k, j), [[1, 4 + j, 2], [1, 5, 6, 2]]) for j in range(3))]
        batches.append(f(ds))
    got = list(data.DevicePrefetcher(batches, dev))
    assert len(got) == 5
    for ref, g in zip(batches, got):
        assert g[0] == ref[0] and g[3][1] == ref[3][1]                            # host objects pass through
        for a, b in ((ref[1], g[1]), (ref[2], g[2]), (ref[3][0], g[3][0]), (ref[4], g[4])):
            assert b.is_cuda and torch.equal(a, b.cpu())
    # ... and with the features handed over once per image (dedup): the same device tensors, expanded on the device;
    # batches of different sizes go through the same (growing) pinned staging buffers
    fd = data.create_collate_fn('caption', max_seq_len=6, dedup=True)
    sizes = (3, 5, 2, 5, 3)
    plain, dd = [], []
    for k, n in enumerate(sizes):
        ds = [(x[0], x[1], x[2], x[3], x[4]) for x in (_img('j%d_%d' % (k, j), [[1, 4 + j, 2], [1, 5, 6, 2], [1, 2]]) for j in range(n))]
        plain.append(f(ds))
        dd.append(fd(ds))
    for ref, g in zip(plain, data.DevicePrefetcher(dd, dev)):
        assert g[1].is_cuda and torch.is_tensor(g[1]) and g[1].shape == ref[1].shape
        assert torch.equal(ref[1], g[1].cpu()) and torch.equal(ref[2], g[2].cpu()) and torch.equal(ref[3][0], g[3][0].cpu())


@pytest.mark.gpu
def test_device_resident_feature_store_yields_the_same_batches():
    """data.DeviceFeatureStore: the features live on the device, a batch is gathered there by row index.  Same batches as
    the dict-backed loaders (caption collate plain and dedup, rl_fact), through DevicePrefetcher and through
    RowGather.to(device) (what Detector.forward's `.to(device)` gets)."""
    dev = torch.device('cuda:0')
    imgs = [_img('k%d' % j, [[1, 4 + j, 2], [1, 5, 6, 2], [1, 2]]) for j in range(7)]
    fns = [x[0] for x in imgs]
    fc, att = {x[0]: x[1] for x in imgs}, {x[0]: x[2] for x in imgs}
    caps, cpts = {x[0]: x[3] for x in imgs}, {x[0]: x[4] for x in imgs}
    sentis = {fn: [3, 4] for fn in fns}
    dfc = data.DeviceFeatureStore.from_arrays(fns, [fc[f] for f in fns], dev, chunk_rows=3)
    datt = data.DeviceFeatureStore.from_store(att, dev)
    assert len(dfc) == 7 and 'k3' in dfc and torch.equal(datt.tensor[datt.index['k2']].cpu(), torch.from_numpy(att['k2']))
    for dedup in (False, True):
        ref = list(data.get_caption_dataloader(fc, att, caps, cpts, 0, 6, 4, 3, shuffle=False))
        got = list(data.DevicePrefetcher(data.get_caption_dataloader(dfc, datt, caps, cpts, 0, 6, 4, 3, shuffle=False,
                                                                     dedup=dedup), dev))
        assert len(ref) == len(got) == 3
        for r, g in zip(ref, got):
            assert r[0] == g[0] and torch.is_tensor(g[1]) and g[1].is_cuda
            assert torch.equal(r[1], g[1].cpu()) and torch.equal(r[2], g[2].cpu()) and torch.equal(r[3][0], g[3][0].cpu())
    import random
    random.seed(3)
    ref = list(data.get_rl_fact_dataloader(fc, att, caps, cpts, sentis, 0, 6, 4, 4, 4, shuffle=False))
    random.seed(3)
    got = list(data.get_rl_fact_dataloader(dfc, datt, caps, cpts, sentis, 0, 6, 4, 4, 4, shuffle=False))
    for r, g in zip(ref, got):
        assert isinstance(g[1], data.RowGather) and g[1].shape == tuple(r[1].shape)
        assert torch.equal(r[1], g[1].to(dev).cpu()) and torch.equal(r[2], g[2].dense().cpu())
        assert torch.equal(r[3][0], g[3][0]) and r[0] == g[0]


# ----------------------------------------------------------------------------- checkpoints
def _cap(V=64):
    c = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.TINY_SETTINGS)
    c.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, synth.TINY_SETTINGS, seed=4).items()})
    return c


def test_checkpoint_roundtrip_and_metadata_checks(tmp_path):
    cap = _cap()
    optim, _, _ = cap.get_optim_criterion(4e-4)
    meta = dict(settings=dict(synth.TINY_SETTINGS), idx2word=synth.make_idx2word(64),
                sentiment_categories=list(synth.SENTIMENT_CATEGORIES), dataset_name='coco', corpus_type='part')
    path = checkpoint.save_xe_checkpoint(str(tmp_path), 3, cap, optim, train_loss=1.5, val_loss=2.25, **meta)
    assert os.path.basename(path).startswith('model_3_1.5000_2.2500_')
    raw = torch.load(path, weights_only=False)
    assert set(raw) == {'epoch', 'model', 'optimizer', 'settings', 'idx2word', 'sentiment_categories',
                        'dataset_name', 'corpus_type'}                              # train_xe.py:241-250
    cap2 = _cap()
    with torch.no_grad():
        cap2.classifier.bias.add_(1.0)
    optim2, _, _ = cap2.get_optim_criterion(1e-3)
    epoch, lr = checkpoint.load_xe_checkpoint(path, cap2, optim2, **meta)
    assert epoch == 3 and lr == 4e-4
    for (k, a), (_, b) in zip(cap.state_dict().items(), cap2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(AssertionError):
        checkpoint.load_xe_checkpoint(path, cap2, None, **dict(meta, dataset_name='flickr30k'))
    checkpoint.write_results(str(tmp_path), 5, [{'image_id': 'a.jpg', 'caption': 'w4 w5'}, {'image_id': 'b.jpg', 'caption': 'w6'}])
    assert json.load(open(tmp_path / 'result_5.json'))[1] == {'image_id': 'b.jpg', 'caption': 'w6'}
    assert open(tmp_path / 'result_5.txt').read() == 'w4 w5\nw6\n'


def test_reference_checkpoint_loads_on_cpu():
    """A file written by the reference (model + torch.optim.Adam state) loads into the build's classes."""
    cap = _cap()
    optim, _, _ = cap.get_optim_criterion(1e-3)
    epoch, lr = checkpoint.load_xe_checkpoint(
        os.path.join(GOLDEN_DIR, 'ref_xe_checkpoint_tiny.pth'), cap, optim, settings=dict(synth.TINY_SETTINGS),
        idx2word=synth.make_idx2word(64), sentiment_categories=list(synth.SENTIMENT_CATEGORIES),
        dataset_name='coco', corpus_type='part')
    assert epoch == 7 and lr == 4e-4
    st = optim.state_dict()['state']
    assert len(st) == 32 and int(st[0]['step']) == 1       # the 8 gate tensors never had a gradient


@pytest.mark.gpu
def test_resume_from_reference_checkpoint_reproduces_its_next_step(golden):
    """Load the reference's checkpoint (after its step 1), run step 2 on the HIP path with the fused
    clamp+Adam, and land where the reference's own second step landed."""
    g = golden('checkpoint')
    dev = torch.device('cuda:0')
    cap = _cap().to(dev).eval()
    optim, xe_crit, da_crit = cap.get_optim_criterion(1e-3)
    checkpoint.load_xe_checkpoint(os.path.join(GOLDEN_DIR, 'ref_xe_checkpoint_tiny.pth'), cap, optim)
    d = synth.make_inputs(6, 64, synth.TINY_SETTINGS, regions=6, seq_len=8, seed=11)
    s = synth.make_inputs(4, 64, synth.TINY_SETTINGS, regions=6, seq_len=8, seed=12)
    t = lambda x, k: torch.from_numpy(np.asarray(x[k])).to(dev)
    pred = cap(t(d, 'fc_feats'), t(d, 'att_feats'), t(d, 'cpt_words'), t(d, 'captions'), t(d, 'senti_labels'), 0.0, mode='xe')
    loss = xe_crit(pred, t(d, 'captions')[:, 1:], d['lengths']) + da_crit(cap.cpt_feats, cap.fc_feats.detach())
    pred2 = cap(t(s, 'captions'), t(s, 'cpt_words'), t(s, 'senti_words'), t(s, 'senti_labels'), 0.0, mode='seq2seq')
    loss = loss + xe_crit(pred2, t(s, 'captions')[:, 1:], s['lengths'])
    np.testing.assert_allclose(float(loss.detach()), g['loss2'][0], rtol=3e-5)
    optim.zero_grad()
    loss.backward()
    clip_gradient(optim, 0.1)
    optim.step()
    for k, v in cap.state_dict().items():
        diff = np.abs(v.cpu().numpy() - g['after2/' + k])
        assert diff.max() <= 2 * 4e-4 * 1.05, k                   # never more than a sign-flipped lr step
        if k.endswith('alpha.bias'):
            continue            # softmax shift invariance: the true gradient is 0, Adam normalises pure rounding noise
        if k.startswith('attention.cont_att.') or k.startswith('attention.senti_att.'):
            # tanh-saturated scorers: gradients of ~1e-7 whose rounding Adam amplifies to a fraction of lr
            assert (diff > 4e-5).mean() < 0.2, (k, float((diff > 4e-5).mean()))
        else:
            assert diff.max() <= 1e-6, (k, float(diff.max()))


def test_synthetic_batches_sorted_by_length_like_the_reference_collates():
    """synth.sort_by_length: longest caption first (stable), every per-row array permuted alike (dataloader.py:17,37)."""
    import numpy as np
    from insenticap_model_amd import synth
    d = synth.make_inputs(12, 64, synth.TINY_SETTINGS, regions=6, seq_len=8, seed=3)
    s = synth.sort_by_length(d)
    assert s['lengths'] == sorted(d['lengths'], reverse=True) and s['lengths'] != d['lengths']
    order = sorted(range(12), key=lambda i: -d['lengths'][i])
    for k in ('captions', 'fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels'):
        np.testing.assert_array_equal(s[k], d[k][order])
    for b, L in enumerate(s['lengths']):
        assert s['captions'][b, L] == 2 and (s['captions'][b, L + 1:] == 0).all()

