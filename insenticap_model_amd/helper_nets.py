"""The two frozen helper networks of the RL step, restated as plain PyTorch-ROCm modules.

They are harness dependencies of the hot path, not part of it:
`Detector` only ever runs them in eval mode without gradients (decoder.py:31-32,83,133-136 and
self_critical/utils.py:120-151), their dense conv / LSTM work goes to MIOpen / rocBLAS through
stock torch ops, and only their *outputs* (sentiment labels, per-token reward weights) enter the
hand-written kernels.  Parameter names and shapes equal the reference's, so its checkpoints load
(train_rl.py:42-53,88-97).

* SentimentDetector            /root/reference/models/sentiment_detector.py:5-64
* SentenceSentimentClassifier  /root/reference/models/sent_senti_cls.py:6-72
"""
import torch
import torch.nn as nn


class SentimentDetector(nn.Module):
    """Image-level sentiment from the [B,h,w,F] feature grid: two 3x3 convs (F -> F/2 -> F/4, no
    non-linearity in between), dropout, ReLU, a 1x1 conv to one map per sentiment, global average
    pooling and `sentiment_fcs_num` small linear layers."""

    def __init__(self, sentiment_categories, settings):
        super().__init__()
        self.sentiment_categories = sentiment_categories
        self.neu_idx = sentiment_categories.index('neutral')
        ch = settings['fc_feat_dim']
        self.convs = nn.Sequential()
        for i in range(settings['sentiment_convs_num']):
            self.convs.add_module('conv_%d' % i, nn.Conv2d(ch, ch // 2, 3, padding=1))
            ch //= 2
        self.convs.add_module('dropout', nn.Dropout(settings['dropout_p']))
        self.convs.add_module('relu', nn.ReLU())
        k = len(sentiment_categories)
        self.senti_conv = nn.Conv2d(ch, k, 1)
        self.global_pool = nn.AdaptiveAvgPool2d(1)
        self.output = nn.Sequential(*[nn.Linear(k, k) for _ in range(settings['sentiment_fcs_num'])])

    def forward(self, features):
        x = self.convs(features.permute(0, 3, 1, 2))
        maps = self.senti_conv(x)                                   # [B,k,h,w]
        logits = self.output(maps.mean(dim=(2, 3)))                 # GAP + FCs
        B, k, h, w = maps.shape
        weighted = torch.bmm(logits.softmax(dim=-1).unsqueeze(1), maps.reshape(B, k, h * w))
        return logits, weighted.reshape(B, h, w)

    @torch.no_grad()
    def sample_device(self, features, senti_threshold=0):
        """`sample` without the category names: (labels [B] int64, sentiment map, max-probabilities), all on the device -
        nothing is read back, so the caller can queue its next step (the beam search takes the labels as a tensor) before
        the host looks at them (`names`)."""
        self.eval()
        logits, maps = self.forward(features)
        scores, labels = logits.softmax(dim=-1).max(dim=-1)
        labels = torch.where(scores < senti_threshold, torch.full_like(labels, self.neu_idx), labels)
        return labels, maps, scores

    def names(self, labels):
        return [self.sentiment_categories[i] for i in labels.tolist()]          # (one read-back, not one per label)

    @torch.no_grad()
    def sample(self, features, senti_threshold=0):
        """-> (labels [B] int64, sentiment map [B,h,w], category names, max-probabilities [B]);
        a maximum probability below the threshold falls back to `neutral`."""
        labels, maps, scores = self.sample_device(features, senti_threshold)
        return labels, maps, self.names(labels), scores

    def get_optim_criterion(self, lr, weight_decay=0):
        return torch.optim.Adam(self.parameters(), lr=lr, weight_decay=weight_decay), nn.CrossEntropyLoss()


class SentenceSentimentClassifier(nn.Module):
    """Sentence-level sentiment: word embedding -> one-layer LSTM -> squeeze-excite style token
    weights (mean over channels of a sigmoid MLP of every hidden state) -> weighted sum -> MLP."""

    def __init__(self, idx2word, sentiment_categories, settings):
        super().__init__()
        self.sentiment_categories = sentiment_categories
        self.pad_id = idx2word.index('<PAD>')
        self.vocab_size = len(idx2word)
        W, H = settings['word_emb_dim'], settings['rnn_hid_dim']
        self.word_embed = nn.Sequential(nn.Embedding(self.vocab_size, W, padding_idx=self.pad_id), nn.ReLU(),
                                        nn.Dropout(settings['dropout_p']))
        self.rnn = nn.LSTM(W, H, bidirectional=False)
        self.drop = nn.Dropout(settings['dropout_p'])
        self.excitation = nn.Sequential(nn.Linear(H, H), nn.ReLU(), nn.Linear(H, H), nn.Sigmoid())
        self.squeeze = nn.AdaptiveAvgPool1d(1)
        self.sent_senti_cls = nn.Sequential(nn.Linear(H, H), nn.ReLU(), nn.Dropout(settings['dropout_p']),
                                            nn.Linear(H, len(sentiment_categories)))

    def forward(self, seqs, lengths):
        """seqs [B,L] int64, lengths: B ints (>= 1), or a device tensor [B]. Returns (logits [B,k], token weights
        [B,max(lengths)] - [B,L] for device lengths).
        Positions at or beyond a row's length contribute nothing (the reference packs the sequences;
        here the padded LSTM outputs are masked instead - the recurrence is causal, so valid positions
        are unaffected by what follows them)."""
        if torch.is_tensor(lengths) and lengths.device == seqs.device and seqs.is_cuda:
            # lengths that live on the device: no host read - the LSTM runs over all L columns (it is causal and the
            # positions at or beyond a row's length are masked below, so the valid positions are what the packed form
            # computes; the token weights come back [B,L] with zeros behind each row's length)
            Lmax, lens = seqs.shape[1], lengths.to(torch.int64)
        else:
            lengths = [int(x) for x in lengths]
            Lmax = max(lengths)
            if seqs.is_cuda:    # (through pinned memory: a pageable copy would hold the host until the stream reaches it)
                lens = torch.tensor(lengths, dtype=torch.int64).pin_memory().to(seqs.device, non_blocking=True)
            else:
                lens = torch.as_tensor(lengths, device=seqs.device)
        x = self.word_embed(seqs[:, :Lmax])                                   # [B,Lmax,W]
        out, _ = self.rnn(x.transpose(0, 1))                                  # time-major LSTM
        out = out.transpose(0, 1)                                             # [B,Lmax,H]
        valid = (torch.arange(Lmax, device=seqs.device).unsqueeze(0) < lens.unsqueeze(1)).to(out.dtype)
        out = self.drop(out * valid.unsqueeze(-1))
        weights = (self.excitation(out) * valid.unsqueeze(-1)).mean(dim=-1)  # [B,Lmax]
        feats = torch.bmm(weights.unsqueeze(1), out).squeeze(1)               # [B,H]
        return self.sent_senti_cls(feats), weights

    @torch.no_grad()
    def sample(self, seqs, lengths):
        self.eval()
        pred, att_weights = self.forward(seqs, lengths)
        result = [int(p.argmax(-1)) for p in pred]
        return result, [self.sentiment_categories[r] for r in result], att_weights

    def get_optim_and_crit(self, lr, weight_decay=0):
        return torch.optim.Adam(self.parameters(), lr=lr, weight_decay=weight_decay), nn.CrossEntropyLoss()
