"""Deterministic synthetic weights and inputs for the caption-decoder hot path.

Everything here is NumPy-only and keyed by explicit integer seeds, so this
container, the GPU box and the golden-fixture generator all see bit-identical
weights and inputs ("build's own generator, NumPy
default_rng(seed) so both boxes agree bit-for-bit").

The parameter names and shapes are the 40 `state_dict` entries of the
reference `Captioner` (/root/reference/models/captioner.py:131-161), which are
part of the checkpoint API (train_xe.py:52,244; train_rl.py:54,71).
"""
import zlib

import numpy as np

SPECIAL_TOKENS = ['<PAD>', '<SOS>', '<EOS>', '<UNK>']  # ids 0..3
SENTIMENT_CATEGORIES = ['positive', 'negative', 'neutral']  # opts.py:25

# opts.py:80-87
DEFAULT_SETTINGS = dict(word_emb_dim=512, fc_feat_dim=2048, att_feat_dim=2048,
                        feat_emb_dim=512, dropout_p=0.5, rnn_hid_dim=512,
                        att_hid_dim=512)

TINY_SETTINGS = dict(word_emb_dim=32, fc_feat_dim=64, att_feat_dim=64,
                     feat_emb_dim=32, dropout_p=0.5, rnn_hid_dim=32,
                     att_hid_dim=32)


def make_idx2word(vocab_size):
    assert vocab_size > len(SPECIAL_TOKENS)
    return SPECIAL_TOKENS + ['w%d' % i for i in range(len(SPECIAL_TOKENS), vocab_size)]


def param_shapes(vocab_size, settings, n_senti=3):
    """The 40 state_dict keys -> shapes, in the reference's registration order."""
    W = settings['word_emb_dim']
    F = settings['fc_feat_dim']
    FA = settings['att_feat_dim']
    E = settings['feat_emb_dim']
    H = settings['rnn_hid_dim']
    A = settings['att_hid_dim']
    V = vocab_size
    s = {}
    s['word_embed.0.weight'] = (V, W)
    s['senti_label_embed.0.weight'] = (n_senti, W)
    s['fc_embed.0.weight'] = (E, F)
    s['fc_embed.0.bias'] = (E,)
    s['cpt2fc.0.weight'] = (E, W)
    s['cpt2fc.0.bias'] = (E,)
    s['att_embed.0.weight'] = (E, FA)
    s['att_embed.0.bias'] = (E,)
    s['att_lstm.weight_ih'] = (4 * H, H + E + W)
    s['att_lstm.weight_hh'] = (4 * H, H)
    s['att_lstm.bias_ih'] = (4 * H,)
    s['att_lstm.bias_hh'] = (4 * H,)
    s['att2att.0.weight'] = (A, E)
    s['att2att.0.bias'] = (A,)
    s['senti2att.0.weight'] = (A, W)
    s['senti2att.0.bias'] = (A,)
    s['attention.cont_att.h2att.weight'] = (A, H)
    s['attention.cont_att.h2att.bias'] = (A,)
    s['attention.cont_att.att_alpha.weight'] = (1, A)
    s['attention.cont_att.att_alpha.bias'] = (1,)
    s['attention.senti_att.h2word.weight'] = (A, H)
    s['attention.senti_att.h2word.bias'] = (A,)
    s['attention.senti_att.label2word.weight'] = (A, W)
    s['attention.senti_att.label2word.bias'] = (A,)
    s['attention.senti_att.word_alpha.weight'] = (1, A)
    s['attention.senti_att.word_alpha.bias'] = (1,)
    s['attention.h2att.weight'] = (A, H)
    s['attention.h2att.bias'] = (A,)
    s['attention.cont2att.weight'] = (A, E)
    s['attention.cont2att.bias'] = (A,)
    s['attention.senti2att.weight'] = (A, E)
    s['attention.senti2att.bias'] = (A,)
    s['attention.att_alpha.weight'] = (1, A)
    s['attention.att_alpha.bias'] = (1,)
    s['lang_lstm.weight_ih'] = (4 * H, E + H)
    s['lang_lstm.weight_hh'] = (4 * H, H)
    s['lang_lstm.bias_ih'] = (4 * H,)
    s['lang_lstm.bias_hh'] = (4 * H,)
    s['classifier.weight'] = (V, H)
    s['classifier.bias'] = (V,)
    return s


# Per-tensor gain over the 1/sqrt(fan_in) uniform bound. With the framework's
# default init the logits are near-uniform and greedy decode emits one token 20
# times (SURVEY 7 "Hard parts"); these gains sharpen the distributions so that
# top-1/top-2 margins are far above fp32 reassociation noise and rows end at
# different lengths.
_GAINS = {
    'word_embed.0.weight': 4.0,
    'classifier.weight': 10.0,
    'classifier.bias': 2.0,
    'att_lstm.weight_ih': 4.0,
    'att_lstm.weight_hh': 2.0,
    'lang_lstm.weight_ih': 4.0,
    'lang_lstm.weight_hh': 2.0,
    'attention.cont_att.att_alpha.weight': 6.0,
    'attention.senti_att.word_alpha.weight': 6.0,
    'attention.att_alpha.weight': 4.0,
}


def make_weights(vocab_size, settings, n_senti=3, seed=0, eos_row_gain=3.0):
    """name -> float32 ndarray; each tensor has its own PCG64 stream keyed by
    (seed, crc32(name)) so adding/removing tensors never shifts the others."""
    out = {}
    shapes = param_shapes(vocab_size, settings, n_senti)
    for name, shape in shapes.items():
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        if name in ('word_embed.0.weight', 'senti_label_embed.0.weight'):
            bound = 1.0  # embedding tables: U(-1,1) * gain
        elif len(shape) == 2:
            bound = 1.0 / np.sqrt(shape[1])
        else:
            # a bias follows the fan-in of its layer's weight matrix
            stem = name.rsplit('.', 1)[0] if name.endswith('.bias') else None
            wname = (stem + '.weight') if stem else name.replace('bias_', 'weight_')
            bound = 1.0 / np.sqrt(shapes[wname][1])
        g = _GAINS.get(name, 1.0)
        out[name] = (rng.uniform(-1.0, 1.0, size=shape) * bound * g).astype(np.float32)
    # nn.Embedding(padding_idx=pad_id) keeps the PAD row at zero (captioner.py:133-135)
    out['word_embed.0.weight'][0, :] = 0.0
    # a higher-variance <EOS> logit makes rows finish at different steps
    out['classifier.weight'][2, :] *= np.float32(eos_row_gain)
    return out


def make_inputs(batch, vocab_size, settings, regions=36, n_cpt=5, n_senti_words=10,
                seq_len=20, n_senti=3, seed=1, grid=None):
    """Synthetic batch of SURVEY 8(d): non-negative U[0,1) features (post-ReLU CNN
    activations), word ids in [4, V), labels in [0, n_senti).

    captions: [B, seq_len+1] with column 0 = <SOS>, random length in
    [seq_len//2, seq_len] tokens incl. the trailing <EOS>, <PAD>=0 after it.
    """
    rng = np.random.default_rng(seed)
    F = settings['fc_feat_dim']
    FA = settings['att_feat_dim']
    d = {}
    d['fc_feats'] = rng.random((batch, F), dtype=np.float32)
    att = rng.random((batch, regions, FA), dtype=np.float32)
    if grid is not None:
        att = att.reshape(batch, grid[0], grid[1], FA)
    d['att_feats'] = att
    d['cpt_words'] = rng.integers(4, vocab_size, size=(batch, n_cpt), dtype=np.int64)
    d['senti_words'] = rng.integers(4, vocab_size, size=(batch, n_senti_words), dtype=np.int64)
    d['senti_labels'] = rng.integers(0, n_senti, size=(batch,), dtype=np.int64)
    caps = np.zeros((batch, seq_len + 1), dtype=np.int64)
    caps[:, 0] = 1
    lengths = rng.integers(max(2, seq_len // 2), seq_len + 1, size=(batch,))
    lengths[0] = seq_len  # XECriterion needs pred.size(1) == max(lengths) (captioner.py:431-440)
    for b in range(batch):
        L = int(lengths[b])
        caps[b, 1:L] = rng.integers(4, vocab_size, size=(L - 1,))
        caps[b, L] = 2
    d['captions'] = caps
    d['lengths'] = [int(x) for x in lengths]
    return d


def sort_by_length(d):
    """make_inputs' batch with the longest caption first (stable), every per-row array permuted alike - the order the
    reference's collates hand a batch over in (dataloader.py:17,37,68,124)."""
    order = sorted(range(len(d['lengths'])), key=lambda i: -d['lengths'][i])
    out = {}
    for k, v in d.items():
        if k == 'lengths':
            out[k] = [v[i] for i in order]
        elif isinstance(v, np.ndarray) and v.shape[:1] == (len(order),):
            out[k] = np.ascontiguousarray(v[order])
        else:
            out[k] = v
    return out


def make_cider_data(n_images, vocab_size, batch, seq_len=20, n_refs=5, seed=7):
    """Synthetic RL-reward inputs: per image `n_refs` ground-truth captions
    [<SOS>, 8..18 Zipf-distributed ids, <EOS>]; a batch of sampled / greedy roll-out rows built by
    perturbing references (so that n-gram overlap - and the reward - is non-trivial), <EOS> at a
    random position and <PAD> after it.  Returns (split_captions, fns, ground_truth, sample, greedy)."""
    rng = np.random.default_rng(seed)
    ranks = np.arange(4, vocab_size)
    prob = 1.0 / (ranks - 3.0)
    prob /= prob.sum()
    captions = {}
    for i in range(n_images):
        caps = []
        for _ in range(n_refs):
            L = int(rng.integers(8, 19))
            caps.append([1] + [int(x) for x in rng.choice(ranks, size=L, p=prob)] + [2])
        captions['img%05d' % i] = caps
    fns = ['img%05d' % int(i) for i in rng.choice(n_images, size=batch, replace=False)]

    def rollout_rows():
        rows = np.zeros((batch, seq_len), dtype=np.int64)
        for b, fn in enumerate(fns):
            ref = captions[fn][int(rng.integers(0, n_refs))][1:-1]
            words = [w if rng.random() < 0.7 else int(rng.choice(ranks, p=prob)) for w in ref]
            words = words[:int(rng.integers(3, seq_len))]
            if len(words) < seq_len and rng.random() < 0.9:
                words = words + [2]                 # most rows end with <EOS>, some run to the limit
            rows[b, :len(words)] = words[:seq_len]
        return rows
    sample, greedy = rollout_rows(), rollout_rows()
    split = {'train': {k: v for k, v in list(captions.items())[:n_images // 2]},
             'val': {k: v for k, v in list(captions.items())[n_images // 2:]}}
    ground_truth = {fn: captions[fn] for fn in fns}
    return split, fns, ground_truth, sample, greedy


# settings keys only the helper nets read (opts.py:89-92)
HELPER_SETTINGS = dict(sentiment_convs_num=2, sentiment_fcs_num=2)


def make_module_weights(shapes, seed, gain=2.0):
    """Deterministic weights for an arbitrary module given {name: shape} (its state_dict layout):
    U(-1,1) * gain / sqrt(fan_in), embeddings U(-1,1) with a zero <PAD> row; one PCG64 stream per name."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(x) for x in shape)
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        if 'embed' in name:
            w = rng.uniform(-1.0, 1.0, size=shape)
            w[0] = 0.0
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else max(shape[0], 1)
            w = rng.uniform(-1.0, 1.0, size=shape) * gain / np.sqrt(fan_in)
        out[name] = w.astype(np.float32)
    return out


def make_rl_batches(n_batches, batch, vocab_size, settings, grid=(2, 3), seq_len=8, seed=40):
    """Fact batches in the reference's rl_fact collate layout (dataloader.py:60-91):
    (fns, fc [B,F], att [B,h,w,F], (caps [B,L], lengths), cpts [B,5], sentis [B,10], ground_truth)
    plus the caption dictionary for the CIDEr-D document frequencies."""
    n_img = n_batches * batch
    split, _, _, _, _ = make_cider_data(n_img, vocab_size, min(batch, n_img), seq_len=seq_len, seed=seed)
    captions = {}
    for v in split.values():
        captions.update(v)
    fns_all = sorted(captions)
    batches = []
    for i in range(n_batches):
        d = make_inputs(batch, vocab_size, settings, regions=grid[0] * grid[1], seq_len=seq_len,
                        seed=seed + 1 + i, grid=grid)
        fns = fns_all[i * batch:(i + 1) * batch]
        gt = {fn: [c[:seq_len + 1] for c in captions[fn]] for fn in fns}
        batches.append((fns, d['fc_feats'], d['att_feats'], (d['captions'], d['lengths']), d['cpt_words'],
                        d['senti_words'], gt))
    return batches, split
