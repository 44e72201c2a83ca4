// Native CIDEr-D (host C++), see include/insenticap_cider.h.  Exact fp64 port of the scorer the
// reference runs in pure Python once per RL iteration (ciderD_scorer.py:120-192), keeping its
// order of operations: n-grams are visited in first-occurrence order (Python dict insertion
// order: all 1-grams by position, then 2-grams, ...), per-order norms and clipped dot products are
// accumulated in that order, the four per-order similarities are summed over the references in
// reference order, then mean over n, / #refs, * 10.
#include <cmath>
#include <cstdint>
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

// counts in first-occurrence order (precook, ciderD_scorer.py:13-28)
struct Cooked {
    std::vector<NGram> keys;
    std::vector<double> tf;
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
    std::unordered_map<NGram, int, NGramHash> index;
    const int64_t L = (int64_t)words.size();
    for (int k = 1; k <= n; ++k)
        for (int64_t i = 0; i + k <= L; ++i) {
            NGram g;
            g.n = k;
            for (int j = 0; j < 4; ++j) g.w[j] = j < k ? words[i + j] : 0;
            auto it = index.find(g);
            if (it == index.end()) {
                index.emplace(g, (int)c.keys.size());
                c.keys.push_back(g);
                c.tf.push_back(1.0);
            } else {
                c.tf[it->second] += 1.0;
            }
        }
}

struct Vec {  // counts2vec result
    std::vector<NGram> keys;
    std::vector<double> val;
    std::unordered_map<NGram, double, NGramHash> map;  // for reference lookups
    double norm[4];
    double length;
};

}  // namespace

struct isc_cider {
    std::unordered_map<NGram, double, NGramHash> df;
    double ref_len;
    int64_t n_imgs, sos, eos;
    int n;
    double sigma;

    void counts2vec(const Cooked &c, Vec &v, bool want_map) const {
        v.keys = c.keys;
        v.val.resize(c.keys.size());
        v.map.clear();
        for (int i = 0; i < 4; ++i) v.norm[i] = 0.0;
        v.length = 0.0;
        for (size_t i = 0; i < c.keys.size(); ++i) {
            const NGram &g = c.keys[i];
            auto it = df.find(g);
            const double d = std::log(std::fmax(1.0, it == df.end() ? 0.0 : it->second));
            const int o = g.n - 1;
            const double x = c.tf[i] * (ref_len - d);
            v.val[i] = x;
            if (want_map) v.map.emplace(g, x);
            v.norm[o] += std::pow(x, 2);
            if (o == 1) v.length += c.tf[i];  // "length" counts bigrams (ciderD_scorer.py:142-143)
        }
        for (int i = 0; i < 4; ++i) v.norm[i] = std::sqrt(v.norm[i]);
    }

    double score_one(const int64_t *hyp, int64_t T, const int64_t *rtok, const int64_t *rcap, int64_t c0,
                     int64_t c1) const {
        std::vector<int64_t> words;
        Cooked ck;
        Vec vh, vr;
        normalise(hyp, T, sos, eos, words);
        precook(words, n, ck);
        counts2vec(ck, vh, false);
        double score[4] = {0, 0, 0, 0};
        for (int64_t c = c0; c < c1; ++c) {
            normalise(rtok + rcap[c], rcap[c + 1] - rcap[c], sos, eos, words);
            precook(words, n, ck);
            counts2vec(ck, vr, true);
            const double delta = vh.length - vr.length;
            double val[4] = {0, 0, 0, 0};
            for (size_t i = 0; i < vh.keys.size(); ++i) {
                const int o = vh.keys[i].n - 1;
                auto it = vr.map.find(vh.keys[i]);
                const double r = it == vr.map.end() ? 0.0 : it->second;
                val[o] += std::fmin(vh.val[i], r) * r;  // clipped (ciderD_scorer.py:164)
            }
            const double pen = std::pow(M_E, -(delta * delta) / (2 * sigma * sigma));
            for (int o = 0; o < n; ++o) {
                if (vh.norm[o] != 0 && vr.norm[o] != 0) val[o] /= (vh.norm[o] * vr.norm[o]);
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
        for (int64_t i = tid; i < n_hyp; i += n_threads)
            scores_out[i] = h->score_one(hyp + i * hyp_stride, T, ref_tokens, ref_cap_off, ref_img_off[i],
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
