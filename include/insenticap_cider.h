/*
 * insenticap_cider.h - C ABI of libinsenticap_cider.so: native CIDEr-D reward for the
 * self-critical RL step (host side of the hot path).
 *
 * Replaces the pure-Python scorer the reference calls once per RL iteration:
 *   self_critical/utils.py:11-21   _array_to_str      (strip <SOS>, cut at first <EOS>, append <EOS>)
 *   self_critical/utils.py:38-53   get_ciderd_scorer  (document frequencies over all GT captions)
 *   self_critical/utils.py:56-83   get_self_critical_reward
 *   self_critical/cider/pyciderevalcap/ciderD/ciderD_scorer.py:13-28,52-64,120-192
 * Token ids are used directly as n-gram symbols (the reference joins them into strings and
 * splits them again). All arithmetic is fp64 in the reference's order of operations, so scores
 * agree to the last bits; scoring is multi-threaded over hypotheses.
 *
 * Captions are passed flattened: `tokens` holds all ids back to back, caption c is
 * tokens[cap_off[c] .. cap_off[c+1]), image i owns captions [img_off[i] .. img_off[i+1]).
 */
#ifndef INSENTICAP_CIDER_H
#define INSENTICAP_CIDER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct isc_cider isc_cider;

/* Build the document-frequency table from the ground-truth captions of all images
 * (CiderScorer.update_df). n = max n-gram order (4), sigma = length-penalty width (6.0). */
isc_cider *isc_cider_create(const int64_t *tokens, const int64_t *cap_off, const int64_t *img_off,
                            int64_t n_imgs, int64_t sos_id, int64_t eos_id, int n, double sigma);
void isc_cider_destroy(isc_cider *h);

/* Number of images and distinct n-grams in the document-frequency table (diagnostics). */
int64_t isc_cider_num_images(const isc_cider *h);
int64_t isc_cider_num_ngrams(const isc_cider *h);

/* CIDEr-D score of n_hyp hypotheses (hyp[i*hyp_stride .. +T), raw roll-out rows) against their
 * own reference captions (flattened like above, hypothesis i owns captions
 * [ref_img_off[i] .. ref_img_off[i+1])).  scores_out[n_hyp] (already x10, mean over refs).
 * Returns 0, or -1 on a null pointer, -2 on a hypothesis without references. */
int isc_cider_score(const isc_cider *h, const int64_t *hyp, int64_t n_hyp, int64_t T,
                    int64_t hyp_stride, const int64_t *ref_tokens, const int64_t *ref_cap_off,
                    const int64_t *ref_img_off, double *scores_out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
