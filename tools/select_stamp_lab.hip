// Stand-alone timing / stamp lab of beam_select_kernel (csrc/pointwise.hip built with ROWS_STAMP 1).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_lab/select_stamp_lab tools/select_stamp_lab.hip
#define ROWS_STAMP 1
#include "../insenticap_model_amd/csrc/pointwise.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
    const int beam = 5, T = 20, V = 10000, n_tile = 250, rows = beam, reps = 200;
    float *pm, *ps, *cv; int *ci; double *sc[2]; long long *last[2], *words[2], *src, *stamp; int *len[2], *done, *live;
    CK(hipMalloc(&pm, rows * n_tile * 4)); CK(hipMalloc(&ps, rows * n_tile * 4));
    CK(hipMalloc(&cv, rows * n_tile * 32)); CK(hipMalloc(&ci, rows * n_tile * 32));
    for (int i = 0; i < 2; ++i) {
        CK(hipMalloc(&sc[i], rows * 8)); CK(hipMalloc(&last[i], rows * 8)); CK(hipMalloc(&words[i], rows * T * 8)); CK(hipMalloc(&len[i], rows * 4));
        CK(hipMemset(sc[i], 0, rows * 8)); CK(hipMemset(words[i], 0, rows * T * 8)); CK(hipMemset(len[i], 0, rows * 4));
    }
    std::vector<long long> hl(rows, 7);
    CK(hipMemcpy(last[0], hl.data(), rows * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(last[1], hl.data(), rows * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&src, rows * 8)); CK(hipMalloc(&done, 4)); CK(hipMalloc(&live, (T + 1) * 4)); CK(hipMalloc(&stamp, 16 * 8 * 8));
    CK(hipMemset(done, 0, 4)); CK(hipMemset(live, 0, (T + 1) * 4));
    std::vector<float> hv(rows * n_tile * 8), hp(rows * n_tile, 1.0f);
    std::vector<int> hi(rows * n_tile * 8);
    for (int r = 0; r < rows; ++r)
        for (int t = 0; t < n_tile; ++t)
            for (int k = 0; k < 8; ++k) {
                hv[(r * n_tile + t) * 8 + k] = (float)(((r * 7 + t * 13) % 101) - k * 3) * 0.01f;
                hi[(r * n_tile + t) * 8 + k] = t * 40 + k;
            }
    CK(hipMemcpy(cv, hv.data(), hv.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ci, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(pm, hp.data(), hp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ps, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
    isc_beam_select_args a = {};
    a.n_img = 1; a.beam = beam; a.T = T; a.t = 1; a.n_tile = n_tile; a.V = V; a.eos_id = 2;
    a.part_max = pm; a.part_sum = ps; a.cand_val = cv; a.cand_idx = ci;
    a.done = done; a.src_row = (int64_t *)src; a.live = live;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int i) {
        a.score_in = sc[i & 1]; a.score_out = sc[(i + 1) & 1]; a.last_in = (const int64_t *)last[0]; a.last_out = (int64_t *)last[1];
        a.words_in = (const int64_t *)words[i & 1]; a.words_out = (int64_t *)words[(i + 1) & 1]; a.len_in = len[0]; a.len_out = len[1];
        return isc_beam_select(&a, st);
    };
    long long *null_stamp = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rows_stamp), &null_stamp, sizeof(null_stamp)));
    for (int i = 0; i < 10; ++i) if (run(i)) return 2;
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run(i);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("select beam=%d n_tile=%d: %.2f us per launch back to back\n", beam, n_tile, 1e3 * ms / reps);
    CK(hipMemset(stamp, 0, 16 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rows_stamp), &stamp, sizeof(stamp)));
    for (int i = 0; i < 3; ++i) { run(i); CK(hipStreamSynchronize(st)); }
    std::vector<long long> hs(16 * 8);
    CK(hipMemcpy(hs.data(), stamp, hs.size() * 8, hipMemcpyDeviceToHost));
    const char *names[7] = {"start", "loads issued", "normaliser", "cands in LDS", "merged", "bookkeeping", "end"};
    long long t0 = hs[0];
    for (int w = 0; w < beam; ++w) t0 = std::min(t0, hs[w * 8]);
    for (int s = 0; s < 7; ++s) {
        printf("  %-14s", names[s]);
        for (int w = 0; w < beam; ++w) printf(" %6lld", hs[w * 8 + s] ? hs[w * 8 + s] - t0 : -1);
        printf("   (x10 ns, per wave)\n");
    }
    printf("  shader cycles across the merge loop (wave 0): %lld\n", hs[7]);
    return 0;
}
