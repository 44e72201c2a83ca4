"""Checkpoint and result files in the reference's formats (SURVEY 5 "Checkpoint / resume", 8(f)-4).

The captioner's 40-tensor `state_dict` and FusedClampAdam's Adam-layout optimizer state make the
files interchangeable with the reference's:
  XE stage  train_xe.py:241-254  {epoch, model, optimizer, settings, idx2word, sentiment_categories,
                                   dataset_name, corpus_type};  resume checks train_xe.py:39-56
  RL stage  train_rl.py:311-325  {epoch, model (Detector state_dict: captioner.* / senti_detector.* /
                                   sent_senti_cls.*), settings, idx2word, max_seq_len, ...}
  results   train_xe.py:230-232  result_<epoch>.json [{image_id, caption}] + result_<epoch>.txt
"""
import json
import os
import time

import torch

_META = ('settings', 'idx2word', 'sentiment_categories', 'dataset_name', 'corpus_type')


def save_xe_checkpoint(directory, epoch, captioner, optimizer, settings, idx2word, sentiment_categories,
                       dataset_name, corpus_type, train_loss=0.0, val_loss=0.0):
    chk = {'epoch': epoch, 'model': captioner.state_dict(), 'optimizer': optimizer.state_dict(),
           'settings': settings, 'idx2word': idx2word, 'sentiment_categories': sentiment_categories,
           'dataset_name': dataset_name, 'corpus_type': corpus_type}
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, 'model_%d_%.4f_%.4f_%s.pth' % (epoch, train_loss, val_loss,
                                                                  time.strftime('%m%d-%H%M')))
    torch.save(chk, path)
    return path


def save_rl_checkpoint(directory, epoch, detector, settings, idx2word, max_seq_len, sentiment_categories,
                       dataset_name, corpus_type):
    chk = {'epoch': epoch, 'model': detector.state_dict(), 'settings': settings, 'idx2word': idx2word,
           'max_seq_len': max_seq_len, 'sentiment_categories': sentiment_categories,
           'dataset_name': dataset_name, 'corpus_type': corpus_type}
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, 'model_%d_%s.pth' % (epoch, time.strftime('%m%d-%H%M')))
    torch.save(chk, path)
    return path


def _check_meta(chk, expected):
    for key in _META:
        if key in expected and expected[key] != chk[key]:
            raise AssertionError('%s and resume model %s are different' % (key, key))


def load_xe_checkpoint(path, captioner, optimizer=None, **expected):
    """Resume as train_xe.py:39-56 does: metadata must match, then model (+ optimizer) state is loaded.
    Returns (epoch, lr)."""
    chk = torch.load(path, map_location=lambda s, l: s, weights_only=False)
    _check_meta(chk, expected)
    captioner.load_state_dict(chk['model'])
    lr = None
    if optimizer is not None and 'optimizer' in chk:
        optimizer.load_state_dict(chk['optimizer'])
        lr = optimizer.param_groups[0]['lr']
    return chk['epoch'], lr


def load_captioner_into_detector(path, detector, **expected):
    """RL warm start (train_rl.py:58-71): the XE checkpoint's `model` goes into `detector.captioner`."""
    chk = torch.load(path, map_location=lambda s, l: s, weights_only=False)
    _check_meta(chk, expected)
    detector.captioner.load_state_dict(chk['model'])
    return chk['epoch']


def write_results(result_dir, epoch, results):
    """results: [{'image_id': fn, 'caption': str}] -> result_<epoch>.json / .txt (train_xe.py:219-232)."""
    os.makedirs(result_dir, exist_ok=True)
    with open(os.path.join(result_dir, 'result_%d.json' % epoch), 'w') as f:
        json.dump(results, f)
    with open(os.path.join(result_dir, 'result_%d.txt' % epoch), 'w') as f:
        f.write(''.join(r['caption'] + '\n' for r in results))
