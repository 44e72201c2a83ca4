// Stand-alone timing / stamp lab of the few-row classifier kernel (csrc/rows.hip built with ROWS_STAMP 1).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/_lab/rows_stamp_lab tools/rows_stamp_lab.hip && tools/_lab/rows_stamp_lab
#define ROWS_STAMP 1
#include "../insenticap_model_amd/csrc/rows.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    setenv("HIP_FORCE_DEV_KERNARG", "1", 0);          // as the package does (kernel arguments in device memory)
    const int M = argc > 1 ? atoi(argv[1]) : 5, V = 10000, K = 512, reps = 200;
    const int TW = isc_rows_stats_tile(V), nt = (V + TW - 1) / TW;
    float *h, *W, *bias, *pmax, *psum, *cv;
    int *pidx, *ci;
    long long *last, *stamp;
    CK(hipMalloc(&h, 8 * K * 4)); CK(hipMalloc(&W, (size_t)V * K * 4)); CK(hipMalloc(&bias, V * 4));
    CK(hipMalloc(&pmax, 8 * nt * 4)); CK(hipMalloc(&psum, 8 * nt * 4)); CK(hipMalloc(&pidx, 8 * nt * 4));
    CK(hipMalloc(&cv, 8 * nt * 8 * 4)); CK(hipMalloc(&ci, 8 * nt * 8 * 4)); CK(hipMalloc(&last, 64));
    CK(hipMalloc(&stamp, (size_t)nt * 16 * 8 * 8));
    std::vector<float> hw((size_t)V * K);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(h, hw.data(), 8 * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hw.data(), V * 4, hipMemcpyHostToDevice));
    CK(hipMemset(last, 0, 64));
    // a second buffer the size of the step's other weights, swept between launches (what a decode step does to the caches)
    float *other; const size_t other_n = 24u << 20;
    CK(hipMalloc(&other, other_n));
    isc_rows_ext x = {};
    x.stats_tile = TW; x.beam = 5; x.cand_val = cv; x.cand_idx = ci; x.last_word = (const int64_t *)last;
    x.pad_id = 0; x.sos_id = 1; x.unk_id = 3; x.mask_special = 1; x.decoding_constraint = 1;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long long *null_stamp = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rows_stamp), &null_stamp, sizeof(null_stamp)));
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 10; ++i) isc_rows_vocab_fwd(h, K, W, K, bias, M, V, K, pmax, psum, pidx, nullptr, 0, &x, st);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) {
            if (mode) CK(hipMemsetAsync(other, 0, other_n, st));
            int rc = isc_rows_vocab_fwd(h, K, W, K, bias, M, V, K, pmax, psum, pidx, nullptr, 0, &x, st);
            if (rc) { printf("rc %d\n", rc); return 1; }
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("vocab M=%d TW=%d tiles=%d %s: %.2f us per iteration\n", M, TW, nt, mode ? "with 24 MB memset between" : "back to back", 1e3 * ms / reps);
    }
    // stamps of one launch
    CK(hipMemset(stamp, 0, (size_t)nt * 16 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rows_stamp), &stamp, sizeof(stamp)));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(other, 0, other_n, st));
        isc_rows_vocab_fwd(h, K, W, K, bias, M, V, K, pmax, psum, pidx, nullptr, 0, &x, st);
        CK(hipStreamSynchronize(st));
    }
    std::vector<long long> hs((size_t)nt * 16 * 8);
    CK(hipMemcpy(hs.data(), stamp, hs.size() * 8, hipMemcpyDeviceToHost));
    long long t0 = 1LL << 62;
    for (int b = 0; b < nt; ++b) for (int w = 0; w < 10; ++w) t0 = std::min(t0, hs[((size_t)b * 16 + w) * 8 + 0]);
    const char *names[7] = {"start", "B1 (staged)", "all issued", "partials", "after B2", "-", "end"};
    for (int role = 0; role < 2; ++role) {
        printf("%s waves: median / max over workgroups of (stamp - first start of the launch), x10 ns\n", role ? "helper" : "streaming");
        for (int s = 0; s < 7; ++s) {
            std::vector<long long> v;
            for (int b = 0; b < nt; ++b)
                for (int w = role ? 8 : 0; w < (role ? 10 : 8); ++w) {
                    const long long t = hs[((size_t)b * 16 + w) * 8 + s];
                    if (t) v.push_back(t - t0);
                }
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("  %-14s median %6lld  p90 %6lld  max %6lld  (n=%zu)\n", names[s], v[v.size() / 2], v[v.size() * 9 / 10], v.back(), v.size());
        }
    }
    {   // shader clock held during the launch: cycles (s_memtime) per 10 ns tick (s_memrealtime), per wave
        std::vector<double> f;
        for (int b = 0; b < nt; ++b)
            for (int w = 0; w < 10; ++w) {
                const long long *q = &hs[((size_t)b * 16 + w) * 8];
                if (q[6] > q[0]) f.push_back((double)q[7] / (double)(q[6] - q[0]) / 10.0);
            }
        std::sort(f.begin(), f.end());
        printf("shader clock: median %.2f GHz (min %.2f, max %.2f)\n", f[f.size() / 2], f.front(), f.back());
    }
    return 0;
}
