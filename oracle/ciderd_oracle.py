"""ORACLE (test infrastructure only): pure-Python restatement of the reference's CIDEr-D reward.

Follows self_critical/utils.py:11-21,38-83 and
self_critical/cider/pyciderevalcap/ciderD/ciderD_scorer.py:13-28,52-64,120-192, written on token-id
lists instead of the reference's space-joined strings. Pinned by tests/golden/cider.npz, which was
produced by the reference's own `get_ciderd_scorer` / `get_self_critical_reward`.
"""
import math
from collections import OrderedDict


def array_to_words(arr, sos, eos):
    """utils.py:11-21 (as a token list): strip a leading <SOS>, cut at the first <EOS>, append <EOS>."""
    arr = [int(x) for x in arr]
    if arr and arr[0] == sos:
        arr = arr[1:]
    out = []
    for w in arr:
        if w == eos:
            break
        out.append(w)
    out.append(eos)
    return out


def ngram_counts(words, n=4):
    """ciderD_scorer.py:13-28; insertion order = first occurrence, 1-grams first."""
    counts = OrderedDict()
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            g = tuple(words[i:i + k])
            counts[g] = counts.get(g, 0) + 1
    return counts


class CiderDOracle:
    def __init__(self, ref_caption_lists, sos, eos, n=4, sigma=6.0):
        self.sos, self.eos, self.n, self.sigma = sos, eos, n, sigma
        self.df = {}
        for caps in ref_caption_lists:            # one document per image (ciderD_scorer.py:52-64)
            seen = set()
            for cap in caps:
                seen.update(ngram_counts(array_to_words(cap, sos, eos), n).keys())
            for g in seen:
                self.df[g] = self.df.get(g, 0.0) + 1.0
        self.ref_len = math.log(float(len(ref_caption_lists)))

    def _vec(self, counts):
        vec = [OrderedDict() for _ in range(self.n)]
        norm = [0.0] * self.n
        length = 0
        for g, tf in counts.items():
            d = math.log(max(1.0, self.df.get(g, 0.0)))
            o = len(g) - 1
            vec[o][g] = float(tf) * (self.ref_len - d)
            norm[o] += vec[o][g] ** 2
            if o == 1:
                length += tf
        return vec, [math.sqrt(x) for x in norm], length

    def score(self, hyp, refs):
        vh, nh, lh = self._vec(ngram_counts(array_to_words(hyp, self.sos, self.eos), self.n))
        score = [0.0] * self.n
        for ref in refs:
            vr, nr, lr = self._vec(ngram_counts(array_to_words(ref, self.sos, self.eos), self.n))
            delta = float(lh - lr)
            for o in range(self.n):
                val = 0.0
                for g, x in vh[o].items():
                    r = vr[o].get(g, 0.0)
                    val += min(x, r) * r
                if nh[o] != 0 and nr[o] != 0:
                    val /= (nh[o] * nr[o])
                val *= math.e ** (-(delta ** 2) / (2 * self.sigma ** 2))
                score[o] += val
        s = sum(score) / self.n
        return s / len(refs) * 10.0

    def self_critical_reward(self, sample, greedy, refs_per_row):
        """[B,T] sample / greedy id matrices -> list of B rewards (sample - greedy)."""
        return [self.score(s, r) - self.score(g, r) for s, g, r in zip(sample, greedy, refs_per_row)]
