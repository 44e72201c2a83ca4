// Native CIDEr-D (host C++), see include/insenticap_cider.h.  Exact fp64 port of the scorer the
// reference runs in pure Python once per RL iteration (ciderD_scorer.py:120-192), keeping its
// order of operations: n-grams are visited in first-occurrence order (Python dict insertion
// order: all 1-grams by position, then 2-grams, ...), per-order norms and clipped dot products are
// accumulated in that order, the four per-order similarities are summed over the references in
// reference order, then mean over n, / #refs, * 10.
#include <cmath>
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/insenticap_cider.h"

namespace {

struct NGram {
    int64_t w[4];
    int n;  // order 1..4
    bool operator==(const NGram &o) const {
        if (n != o.n) return false;
        for (int i = 0; i < n; ++i)
            if (w[i] != o.w[i]) return false;
        return true;
    }
};
struct NGramHash {
    size_t operator()(const NGram &g) const {
        uint64_t h = 0x9E3779B97F4A7C15ull * (uint64_t)g.n;
        for (int i = 0; i < g.n; ++i) {
            h ^= (uint64_t)g.w[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
            h *= 0xBF58476D1CE4E5B9ull;
        }
        return (size_t)(h ^ (h >> 31));
    }
};

// Small open-addressing index over the n-grams of ONE caption (<= a few hundred entries); buffers
// are reused across captions so the scoring loop performs no heap allocation.
struct LocalIndex {
    std::vector<int> slot;   // -1 = empty, else index into keys
    size_t mask = 0;
    void reset(size_t n_items) {
        size_t cap = 64;
        while (cap < 4 * n_items) cap <<= 1;
        if (slot.size() != cap) slot.assign(cap, -1);
        else std::fill(slot.begin(), slot.end(), -1);
        mask = cap - 1;
    }
    // returns the position of g in keys, or -1 (and, if `insert_as` >= 0, records it)
    int find(const std::vector<NGram> &keys, const NGram &g, int insert_as) {
        size_t h = NGramHash()(g) & mask;
        for (;;) {
            const int s = slot[h];
            if (s < 0) {
                if (insert_as >= 0) slot[h] = insert_as;
                return -1;
            }
            if (keys[s] == g) return s;
            h = (h + 1) & mask;
        }
    }
};

// counts in first-occurrence order (precook, ciderD_scorer.py:13-28)
struct Cooked {
    std::vector<NGram> keys;
    std::vector<double> tf;
    LocalIndex index;
};

// _array_to_str (self_critical/utils.py:11-21): drop a leading <SOS>, stop at the first <EOS>, append <EOS>
static void normalise(const int64_t *arr, int64_t len, int64_t sos, int64_t eos, std::vector<int64_t> &out) {
    out.clear();
    int64_t i = 0;
    if (len > 0 && arr[0] == sos) i = 1;
    for (; i < len; ++i) {
        if (arr[i] == eos) break;
        out.push_back(arr[i]);
    }
    out.push_back(eos);
}

static void precook(const std::vector<int64_t> &words, int n, Cooked &c) {
    c.keys.clear();
    c.tf.clear();
    const int64_t L = (int64_t)words.size();
    c.index.reset((size_t)(n * L));
    for (int k = 1; k <= n; ++k)
        for (int64_t i = 0; i + k <= L; ++i) {
            NGram g;
            g.n = k;
            for (int j = 0; j < 4; ++j) g.w[j] = j < k ? words[i + j] : 0;
            const int at = c.index.find(c.keys, g, (int)c.keys.size());
            if (at < 0) {
                c.keys.push_back(g);
                c.tf.push_back(1.0);
            } else {
                c.tf[at] += 1.0;
            }
        }
}

struct Vec {  // counts2vec result; keys / index live in the Cooked it was built from
    std::vector<double> val;
    double norm[4];
    double length;
};

struct Scratch {  // per-thread buffers
    std::vector<int64_t> words;
    Cooked hyp, ref;
    Vec vh, vr;
};

}  // namespace

struct isc_cider {
    std::unordered_map<NGram, double, NGramHash> df;
    double ref_len;
    int64_t n_imgs, sos, eos;
    int n;
    double sigma;

    void counts2vec(const Cooked &c, Vec &v) const {
        v.val.resize(c.keys.size());
        for (int i = 0; i < 4; ++i) v.norm[i] = 0.0;
        v.length = 0.0;
        for (size_t i = 0; i < c.keys.size(); ++i) {
            const NGram &g = c.keys[i];
            auto it = df.find(g);
            const double d = std::log(std::fmax(1.0, it == df.end() ? 0.0 : it->second));
            const int o = g.n - 1;
            const double x = c.tf[i] * (ref_len - d);
            v.val[i] = x;
            v.norm[o] += std::pow(x, 2);
            if (o == 1) v.length += c.tf[i];  // "length" counts bigrams (ciderD_scorer.py:142-143)
        }
        for (int i = 0; i < 4; ++i) v.norm[i] = std::sqrt(v.norm[i]);
    }

    double score_one(Scratch &S, const int64_t *hyp, int64_t T, const int64_t *rtok, const int64_t *rcap,
                     int64_t c0, int64_t c1) const {
        normalise(hyp, T, sos, eos, S.words);
        precook(S.words, n, S.hyp);
        counts2vec(S.hyp, S.vh);
        double score[4] = {0, 0, 0, 0};
        for (int64_t c = c0; c < c1; ++c) {
            normalise(rtok + rcap[c], rcap[c + 1] - rcap[c], sos, eos, S.words);
            precook(S.words, n, S.ref);
            counts2vec(S.ref, S.vr);
            const double delta = S.vh.length - S.vr.length;
            double val[4] = {0, 0, 0, 0};
            for (size_t i = 0; i < S.hyp.keys.size(); ++i) {
                const int o = S.hyp.keys[i].n - 1;
                const int at = S.ref.index.find(S.ref.keys, S.hyp.keys[i], -1);
                const double r = at < 0 ? 0.0 : S.vr.val[at];
                val[o] += std::fmin(S.vh.val[i], r) * r;  // clipped (ciderD_scorer.py:164)
            }
            const double pen = std::pow(M_E, -(delta * delta) / (2 * sigma * sigma));
            for (int o = 0; o < n; ++o) {
                if (S.vh.norm[o] != 0 && S.vr.norm[o] != 0) val[o] /= (S.vh.norm[o] * S.vr.norm[o]);
                val[o] *= pen;
                score[o] += val[o];
            }
        }
        double s = 0.0;
        for (int o = 0; o < n; ++o) s += score[o];
        s /= (double)n;
        s /= (double)(c1 - c0);
        return s * 10.0;
    }
};

extern "C" isc_cider *isc_cider_create(const int64_t *tokens, const int64_t *cap_off, const int64_t *img_off,
                                       int64_t n_imgs, int64_t sos_id, int64_t eos_id, int n, double sigma) {
    if (!tokens || !cap_off || !img_off || n_imgs <= 0 || n < 1 || n > 4) return nullptr;
    isc_cider *h = new isc_cider();
    h->n_imgs = n_imgs; h->sos = sos_id; h->eos = eos_id; h->n = n; h->sigma = sigma;
    h->ref_len = std::log((double)n_imgs);
    std::vector<int64_t> words;
    Cooked ck;
    for (int64_t i = 0; i < n_imgs; ++i) {
        // every n-gram counts once per image, whichever of its references contain it (compute_doc_freq)
        std::unordered_map<NGram, char, NGramHash> seen;
        for (int64_t c = img_off[i]; c < img_off[i + 1]; ++c) {
            normalise(tokens + cap_off[c], cap_off[c + 1] - cap_off[c], sos_id, eos_id, words);
            precook(words, n, ck);
            for (const NGram &g : ck.keys) seen.emplace(g, 1);
        }
        for (auto &kv : seen) h->df[kv.first] += 1.0;
    }
    return h;
}

extern "C" void isc_cider_destroy(isc_cider *h) { delete h; }
extern "C" int64_t isc_cider_num_images(const isc_cider *h) { return h ? h->n_imgs : 0; }
extern "C" int64_t isc_cider_num_ngrams(const isc_cider *h) { return h ? (int64_t)h->df.size() : 0; }

extern "C" int isc_cider_score(const isc_cider *h, const int64_t *hyp, int64_t n_hyp, int64_t T,
                               int64_t hyp_stride, const int64_t *ref_tokens, const int64_t *ref_cap_off,
                               const int64_t *ref_img_off, double *scores_out, int n_threads) {
    if (!h || !hyp || !ref_tokens || !ref_cap_off || !ref_img_off || !scores_out) return -1;
    for (int64_t i = 0; i < n_hyp; ++i)
        if (ref_img_off[i + 1] <= ref_img_off[i]) return -2;
    if (n_threads < 1) n_threads = 1;
    if ((int64_t)n_threads > n_hyp) n_threads = (int)n_hyp;
    auto work = [&](int tid) {
        Scratch S;
        for (int64_t i = tid; i < n_hyp; i += n_threads)
            scores_out[i] = h->score_one(S, hyp + i * hyp_stride, T, ref_tokens, ref_cap_off, ref_img_off[i],
                                         ref_img_off[i + 1]);
    };
    if (n_threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
        for (auto &t : th) t.join();
    }
    return 0;
}
