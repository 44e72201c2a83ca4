// Grouped, K-segmented GEMM (NT forward, NN / TN backward) with fused epilogues, on two engines:
//   * fp32 MFMA (v_mfma_f32_32x32x2_f32): every layout, every size (tile kernels below);
//   * split-f16 (v_mfma_f32_32x32x16_f16 on hi/lo f16 planes of the fp32 operands, fp32 accumulators): forward and
//     backward launches alike - the large tiles gemm_h3_kernel (128 x 128) / gemm_h3x_kernel (256 x 128) /
//     gemm_h3m_kernel (64 x 128), the skinny tiles gemm_h3s_kernel (32 x 32, 64 x 64, 32 x 128 vocabulary; K split over
//     the waves of a workgroup and, for long contractions, over workgroups), h3_split_kernel for the weight planes
//     (and both operands of the TN layout); dispatch in try_h3s / try_h3 / try_h3_tn further down.
//
//   acc[M,N] = sum_s A_s[M,K_s] * W_s[N,K_s]^T          (s = up to 4 K-segments)
//
// K-segments replace the reference's torch.cat([...],1) in front of nn.LSTMCell / nn.Linear
// (captioner.py:174,180 and the three summed projections of :107-110): no concatenated
// activation buffer and no packed weight copy is ever materialised - every segment reads the
// caller's tensors where they live.  Up to 3 independent problems share one launch.
//
// Tiling for gfx950: 256-thread workgroup = 4 wavefronts of 64; each wave owns a
// 32 x (32*TN) accumulator built from 32x32x2 fp32 MFMAs (exact fp32 FMA chains, so the
// result matches an fp32 CPU GEMM to rounding-order noise - bf16/TF32-like paths would
// break the 1e-4 log-prob parity bound).  Two tile shapes:
//   L: 128 x 128 (4 waves stacked in M, TN=4)  - batch-sized problems
//   S:  32 x 128 (4 waves side by side in N)   - small-M problems (beam rows, B<=128)
// K is consumed in 32-wide chunks, global -> registers -> LDS (double buffered, rows padded
// to 36 floats so that the ds_read_b128 fragment reads are bank-conflict free), one
// barrier per chunk.  The finished tile is staged through LDS once so that every epilogue
// sees (row, col) coordinates and writes full, coalesced rows.
#include <atomic>
#include <mutex>
#include <type_traits>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BK 32
#define LDT 36  // padded LDS row (floats): 36*r mod 64 is a distinct multiple of 4 for 16 rows

enum { EPI_LINEAR = 0, EPI_LSTM = 1, EPI_VOCAB = 2 };

struct DevSeg {
    const float *A;
    const float *W;
    int lda, ldw, K, pad;
    const _Float16 *A_hi, *A_lo;      // caller-provided planes of A ([M,K] contiguous) or null
};

// Plane layout (split-f16 operands): one buffer per [rows, K] tensor, K % 32 == 0; per row and per 32-deep k-block
// the 32 hi values are followed by the 32 lo values, so the 64 bytes of hi and the 64 bytes of lo that one chunk
// consumes form one 128-byte line:  hi(row, k) at base[row * 2K + (k >> 5) * 64 + (k & 31)],  lo = the same + 32.
// The `lo` pointers carried around are `hi + 32`.
__host__ __device__ __forceinline__ long long plane_index(long long row, int k, int K) {
    return row * 2 * K + (k >> 5) * 64 + (k & 31);
}
struct DevASeg {                      // one K-segment of the split-f16 A operand
    const _Float16 *hi, *lo;
    int ld, K;                        // ld = row stride in halfs = 2 * K of the tensor the planes belong to
};

struct DevProb {
    DevSeg seg[ISC_MAX_SEG];
    int nseg, M, N, relu;
    const float *bias0, *bias1, *bias2;
    const uint8_t *mask;
    float mask_scale;
    int ldc, accumulate;
    float *C, *C_pre;
    // lstm
    const float *c_prev;
    float *h_out, *c_out, *gates_out, *hdrop;
    const uint8_t *hmask;
    const float *pre, *tab;
    const int64_t *tab_ids;
    long long tab_ids_stride;
    int H, m_fastest;
    // vocab
    float *pmax, *psum;
    int *pidx;
    long long ld_logits;
    int tiles_m, tiles_n, tile_start, ntile_total, grp_n;
    // split-K (small-M problems): workgroup (tile, ks) contracts chunk range ks and writes its raw
    // partial tile to slab[ks]; a second kernel sums the slabs in order and applies the epilogue
    int ksplit;
    float *slab;
    long long slab_stride;
    // split-f16 path (gemm_h3_kernel): W planes of the K-concatenated weights [N, Kp]; A planes by segment (either one
    // packed segment built by the split kernel, or the caller's per-tensor planes)
    const _Float16 *Wh, *Wl;
    DevASeg ap[ISC_MAX_SEG];
    int nap, Kp;
    _Float16 *h_hi, *h_lo;            // LSTM epilogue: planes of h_out for the consumers
};

struct DevLaunch {
    DevProb p[3];
    int nprob, total_tiles;
    const int *gate;            // optional: *gate == 0 -> the launch returns at once (isc_set_stream_gate; forward entry points)
};

// Operand layouts. "k-minor": element (row, k) at ptr[row*ld + k] (contraction index contiguous;
// activations [M,K] and nn.Linear weights [N,K] in the forward pass).  "k-major": element (row, k)
// at ptr[k*ld + row] - the operand is read through its transpose, which is how the backward pass
// consumes them: dX = dY * W uses W [K=N_out, N=K_in] as a k-major B operand ("NN"), and
// dW = dY^T * X contracts over the batch rows with both operands k-major ("TN").
template <int ROWS, bool KM>
struct Tile {
    // LDS footprint (floats) of one [ROWS x 32] operand tile
    static constexpr int SIZE = KM ? BK * (ROWS + 4) : ROWS * LDT;
    static constexpr int NLD = ROWS / 32;  // float4 loads per thread per chunk
};

// Logical tile -> (problem, tm, tn, k-split slice).  XCD-aware order: blocks with equal blockIdx%8 share
// an XCD (and its L2); give each XCD one contiguous run of logical tiles so that neighbours re-use
// A / W panels from L2.
__device__ __forceinline__ void map_tile(const DevLaunch &L, int &pi, int &tm, int &tn, int &ks, int &ksplit) {
    int logical;
    {
        const int nt = L.total_tiles, bid = blockIdx.x;
        const int q = nt >> 3, r = nt & 7, xcd = bid & 7, j = bid >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    pi = 0;
    if (L.nprob > 1 && logical >= L.p[1].tile_start) pi = 1;
    if (L.nprob > 2 && logical >= L.p[2].tile_start) pi = 2;
    const DevProb &P = L.p[pi];
    int t = logical - P.tile_start;
    ksplit = P.ksplit > 1 ? P.ksplit : 1;
    ks = t % ksplit;
    t /= ksplit;
    if (P.m_fastest) {
        // weights are the larger operand: split the N tiles into 8 groups (one per XCD run) so that a
        // group's weight slice stays resident in that XCD's 4 MB L2 while the activations stream
        // through once: inside a group tm is the outer index and tn the inner one.
        const int per_grp = P.grp_n * P.tiles_m;
        const int g = t / per_grp, r = t - g * per_grp;
        const int rest = P.tiles_n - g * P.grp_n;
        const int gn = rest < P.grp_n ? rest : P.grp_n;
        tm = r / gn;
        tn = g * P.grp_n + (r - tm * gn);
    } else {
        tm = t / P.tiles_n;
        tn = t % P.tiles_n;
    }
}

// ---- epilogues, shared by the tile shapes ---------------------------------------------------------
// Vocabulary epilogue of one 32 x (32*TN) accumulator fragment, straight from the registers (no LDS
// round trip).  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5), so
// one register index r is one output row per half-wave and the row's 32*TN columns of this wave sit
// on the 32 lanes of that half.  Per row: max / arg-max / sum exp(x - max) by 5-step butterflies
// inside the half-wave (4 DPP steps inside the 16-lane rows + one swizzle across them).  frow0 = tile-relative first row of the
// fragment, fcol0 = tile-relative first column.  WN == 1 writes the tile statistics directly,
// otherwise they go to smx/ssm/six[wn][BM] for the cross-wave combine.
template <int TN, int WN, int BM>
__device__ __forceinline__ void epi_vocab_frag(const DevProb &P, f32x16 (&acc)[TN], int frow0, int fcol0, int wn,
                                               int lane, int row0, int col0, int tn, float *smem) {
    const int M = P.M, N = P.N;
    float bv[TN];
    bool cok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gn = col0 + fcol0 + j * 32 + (lane & 31);
        cok[j] = gn < N;
        bv[j] = cok[j] ? P.bias0[gn] : 0.f;
    }
    float *smx = smem;                    // [WN][BM] cross-wave combine (WN > 1 only)
    float *ssm = smem + WN * BM;
    int *six = reinterpret_cast<int *>(smem + 2 * WN * BM);
    // Phased over the fragment's 16 rows, every phase straight-line code: the 16 dependent reduction
    // chains interleave instead of running one after the other between per-row branches.
    float mx[16], sm[16];
    int ix[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        mx[r] = -INFINITY;
        ix[r] = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            acc[j][r] = cok[j] ? acc[j][r] + bv[j] : -INFINITY;      // acc now holds the logits
            const int gn = col0 + fcol0 + j * 32 + (lane & 31);
            if (acc[j][r] > mx[r]) { mx[r] = acc[j][r]; ix[r] = gn; }   // j ascending => smaller column wins ties
        }
    }
    if (P.C) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = row0 + frow0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (gm < M) {
                float *crow = P.C + (long long)gm * P.ld_logits + col0 + fcol0 + (lane & 31);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (cok[j]) crow[j * 32] = acc[j][r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) half_argmax(mx[r], ix[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sm[r] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) sm[r] += cok[j] ? __expf(acc[j][r] - mx[r]) : 0.f;   // padded columns add 0
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sm[r] = half_sum(sm[r]);
    if ((lane & 31) == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = frow0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int gm = row0 + row;
            if constexpr (WN == 1) {
                if (gm < M) {
                    const long long o = (long long)gm * P.ntile_total + tn;
                    P.pmax[o] = mx[r];
                    P.psum[o] = sm[r];
                    P.pidx[o] = ix[r];
                }
            } else {
                smx[wn * BM + row] = mx[r];
                ssm[wn * BM + row] = sm[r];
                six[wn * BM + row] = ix[r];
            }
        }
    }
}

// accumulator fragment -> Cs[BM][LDC] (row-major tile image in LDS)
template <int TN>
__device__ __forceinline__ void epi_stage_frag(float *Cs, int LDC, f32x16 (&acc)[TN], int frow0, int fcol0, int lane) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = frow0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = fcol0 + j * 32 + (lane & 31);
            Cs[row * LDC + col] = acc[j][r];
        }
}

template <int BM, int BN>
__device__ __forceinline__ void epi_linear_tile(const DevProb &P, const float *Cs, int row0, int col0, int tid,
                                                int ks, int ksplit) {
    constexpr int LDC = BN + 4;
    const int M = P.M, N = P.N;
    if (ksplit > 1) {      // raw partial tile -> slab[ks] ([M,N], ld = N, N % 4 == 0)
        float *slab = P.slab + (long long)ks * P.slab_stride;
        for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
            const int row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
            const int gm = row0 + row, gn = col0 + c4;
            if (gm < M && gn < N)
                *reinterpret_cast<float4 *>(slab + (long long)gm * N + gn) =
                    *reinterpret_cast<const float4 *>(Cs + row * LDC + c4);
        }
        return;
    }
    const bool vec = (P.ldc & 3) == 0;
    for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
        const int row = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
        const int gm = row0 + row, gn = col0 + c4;
        if (gm >= M || gn >= N) continue;
        float4 v = *reinterpret_cast<const float4 *>(Cs + row * LDC + c4);
        float o[4] = {v.x, v.y, v.z, v.w}, pre[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = gn + e;
            if (n < N) {
                if (P.bias0) o[e] += P.bias0[n];
                if (P.bias1) o[e] += P.bias1[n];
                if (P.bias2) o[e] += P.bias2[n];
                if (P.accumulate) o[e] += P.C[(long long)gm * P.ldc + n];
                if (P.relu) o[e] = isc_relu(o[e]);
                pre[e] = o[e];
                if (P.mask) o[e] = o[e] * (float)P.mask[(long long)gm * N + n] * P.mask_scale;
            }
        }
        float *dst = P.C + (long long)gm * P.ldc + gn;
        if (vec && gn + 3 < N) {
            *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            if (P.C_pre)
                *reinterpret_cast<float4 *>(P.C_pre + (long long)gm * P.ldc + gn) =
                    make_float4(pre[0], pre[1], pre[2], pre[3]);
        } else {
            for (int e = 0; e < 4 && gn + e < N; ++e) {
                dst[e] = o[e];
                if (P.C_pre) P.C_pre[(long long)gm * P.ldc + gn + e] = pre[e];
            }
        }
    }
}

// LSTM cell update from the four gate pre-activations of NE (row, unit) elements, gate order i, f, g, o.
// The epilogue's global operands - hoisted `pre` term, embedding-table row (behind its token id),
// c_prev - are fetched for all NE elements BEFORE any arithmetic: issued element by element inside the
// compute loop they formed NE serial dependent round trips (att-LSTM at B=4096: +24 us per launch).
// Per element:  g += (b_ih + b_hh);  g += pre;  g += table row   (same order in every kernel).
template <int NE>
__device__ __forceinline__ void lstm_cells(const DevProb &P, const int (&gm)[NE], int unit, const bool (&ok)[NE],
                                           float (&g)[NE][4]) {
    const int H = P.H;
    const bool has_b = P.bias0 != nullptr, has_pre = P.pre != nullptr, has_tab = P.tab != nullptr;
    float q[NE][4], t[NE][4], b[4], cp[NE];
    long long tok[NE];
    if (has_tab) {
#pragma unroll
        for (int e = 0; e < NE; ++e) tok[e] = ok[e] ? P.tab_ids[(long long)gm[e] * P.tab_ids_stride] : 0;
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) cp[e] = ok[e] ? P.c_prev[(long long)gm[e] * H + unit] : 0.f;
    if (has_pre) {
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k) q[e][k] = ok[e] ? P.pre[(long long)gm[e] * 4 * H + k * H + unit] : 0.f;
    }
    if (has_tab) {
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[e][k] = ok[e] ? P.tab[tok[e] * 4 * H + k * H + unit] : 0.f;
    }
    if (has_b) {
#pragma unroll
        for (int k = 0; k < 4; ++k) b[k] = P.bias0[k * H + unit] + P.bias1[k * H + unit];
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (has_b) g[e][k] += b[k];
            if (has_pre) g[e][k] += q[e][k];
            if (has_tab) g[e][k] += t[e][k];
        }
        const float gi = isc_sigmoid(g[e][0]), gf = isc_sigmoid(g[e][1]), gg = isc_tanh(g[e][2]), go = isc_sigmoid(g[e][3]);
        const float c2 = gf * cp[e] + gi * gg;
        const float h2 = go * isc_tanh(c2);
        if (ok[e]) {
            const long long o = (long long)gm[e] * H + unit;
            P.c_out[o] = c2;
            P.h_out[o] = h2;
            if (P.h_hi) {
                const _Float16 hh = (_Float16)h2;
                const long long po = plane_index(gm[e], unit, H);
                P.h_hi[po] = hh;
                P.h_lo[po] = (_Float16)((h2 - (float)hh) * 2048.f);
            }
            if (P.hmask) P.hdrop[o] = h2 * (float)P.hmask[o] * P.mask_scale;
            if (P.gates_out) {
                float *go_ = P.gates_out + (long long)gm[e] * 4 * H + unit;
                go_[0] = gi; go_[H] = gf; go_[2 * H] = gg; go_[3 * H] = go;
            }
        }
    }
}

// gate-interleaved tile (columns = 4 gates x 32 units) in LDS -> cells of units tn*32 .. +32.  A thread's
// elements are rows tid/32 + 8*i of ONE unit (tid % 32).
template <int BM, int BN>
__device__ __forceinline__ void epi_lstm_tile(const DevProb &P, const float *Cs, int row0, int tn, int tid) {
    constexpr int LDC = BN + 4, NE = 4;
    static_assert(BM * 32 % (256 * NE) == 0, "tile rows per thread must be a multiple of the batch");
    const int u = tid & 31;
    for (int base = tid; base < BM * 32; base += 256 * NE) {
        int gm[NE];
        bool ok[NE];
        float g[NE][4];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int row = (base + 256 * e) >> 5;
            gm[e] = row0 + row;
            ok[e] = gm[e] < P.M;
            const float *cr = Cs + row * LDC + u;
            g[e][0] = cr[0]; g[e][1] = cr[32]; g[e][2] = cr[64]; g[e][3] = cr[96];
        }
        lstm_cells<NE>(P, gm, tn * 32 + u, ok, g);
    }
}

// The same from a 32 x 128 accumulator fragment (fragment j = gate j, lane&31 = unit; C/D row map): used by
// the XL tile, whose lone workgroup per CU has nobody to hide an LDS round trip behind.
__device__ __forceinline__ void epi_lstm_frag(const DevProb &P, f32x16 (&acc)[4], int frow0, int lane, int row0, int tn) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int gm[8];
        bool ok[8];
        float g[8][4];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = half * 8 + e;
            gm[e] = row0 + frow0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            ok[e] = gm[e] < P.M;
#pragma unroll
            for (int k = 0; k < 4; ++k) g[e][k] = acc[k][r];
        }
        lstm_cells<8>(P, gm, tn * 32 + (lane & 31), ok, g);
    }
}

// Linear epilogue of TN sub-tiles of 32 columns starting at tile column fcol0.  EDGE: the fragment touches the
// matrix border (per-element predicates); FEAT: accumulate / pre-activation copy / keep-mask in play.  Interior
// fragments run without per-element branches whatever the features.
template <int TN, bool EDGE, bool FEAT, bool RELU>
__device__ __forceinline__ void epi_linear_frag_impl(const DevProb &P, f32x16 (&acc)[TN], int frow0, int fcol0, int lane,
                                                     int row0, int col0) {
    const int M = P.M, N = P.N;
    float b0[TN], b1[TN], b2[TN];
    bool cok[TN];
    const bool h0 = P.bias0 != nullptr, h1 = P.bias1 != nullptr, h2 = P.bias2 != nullptr;
    const bool f_acc = FEAT && P.accumulate, f_pre = FEAT && P.C_pre != nullptr, f_mask = FEAT && P.mask != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gn = col0 + fcol0 + j * 32 + (lane & 31);
        cok[j] = !EDGE || gn < N;
        b0[j] = (h0 && cok[j]) ? P.bias0[gn] : 0.f;
        b1[j] = (h1 && cok[j]) ? P.bias1[gn] : 0.f;
        b2[j] = (h2 && cok[j]) ? P.bias2[gn] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = row0 + frow0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (EDGE && gm >= M) continue;
        float *crow = P.C + (long long)gm * P.ldc + col0 + fcol0 + (lane & 31);
        float prev[TN];
        if (f_acc) {                                   // all TN loads of the row before any arithmetic
#pragma unroll
            for (int j = 0; j < TN; ++j) prev[j] = (!EDGE || cok[j]) ? crow[j * 32] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (EDGE && !cok[j]) continue;
            float o = acc[j][r];
            if (h0) o += b0[j];
            if (h1) o += b1[j];
            if (h2) o += b2[j];
            if (f_acc) o += prev[j];
            if (RELU) o = isc_relu(o);
            if constexpr (FEAT) {
                const int gn = col0 + fcol0 + j * 32 + (lane & 31);
                if (f_pre) P.C_pre[(long long)gm * P.ldc + gn] = o;
                if (f_mask) o = o * (float)P.mask[(long long)gm * N + gn] * P.mask_scale;
            }
            crow[j * 32] = o;
        }
    }
}

template <int TN>
__device__ __forceinline__ void epi_linear_frag(const DevProb &P, f32x16 (&acc)[TN], int frow0, int fcol0, int lane,
                                                int row0, int col0) {
    const bool feat = P.accumulate || P.C_pre || P.mask;
    const bool edge = !(row0 + frow0 + 32 <= P.M && col0 + fcol0 + 32 * TN <= P.N);
    const bool relu = P.relu != 0;
#define ISC_EPI_CASE(E, F, R) epi_linear_frag_impl<TN, E, F, R>(P, acc, frow0, fcol0, lane, row0, col0)
    if (edge) {
        if (relu) ISC_EPI_CASE(true, true, true); else ISC_EPI_CASE(true, true, false);
    } else if (feat) {
        if (relu) ISC_EPI_CASE(false, true, true); else ISC_EPI_CASE(false, true, false);
    } else {
        if (relu) ISC_EPI_CASE(false, false, true); else ISC_EPI_CASE(false, false, false);
    }
#undef ISC_EPI_CASE
}

template <int WM, int WN, int TN, int EPI, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    rows_kernarg_warm<ROWS_KERNARG_LINES(DevLaunch)>();      // (one batch of scalar loads for the launch descriptor: common.h)
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * TN * WN;
    constexpr int LDC = BN + 4;
    using TA = Tile<BM, AKM>;
    using TB = Tile<BN, BKM>;
    constexpr int A_LD = TA::NLD, B_LD = TB::NLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                     // [2][TA::SIZE]
    float *Bs = smem + 2 * TA::SIZE;      // [2][TB::SIZE]
    float *Cs = smem;                     // [BM][LDC] (after the K loop)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    int pi, tm, tn, ks, ksplit;
    map_tile(L, pi, tm, tn, ks, ksplit);
    const DevProb &P = L.p[pi];
    const int M = P.M, N = P.N;
    const int row0 = tm * BM, col0 = tn * BN;

    // ---- global-load coordinates
    // k-minor tile: thread -> (row = tid/8 + 32*i, k = 4*(tid%8)); k-major tile of ROWS columns:
    // thread -> (k-row = tid/(ROWS/4) + (1024/ROWS)*i, column = 4*(tid % (ROWS/4)))
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    constexpr int A_TPR = BM / 4, B_TPR = BN / 4;           // threads per k-row (k-major)
    const int akr = tid / A_TPR, akc = (tid % A_TPR) * 4;
    const int bkr = tid / B_TPR, bkc = (tid % B_TPR) * 4;
    long long wrow[B_LD];   // k-minor B: weight row of tile row n (LSTM tiles interleave the 4 gates)
    bool wok[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int n = lr + 32 * i;
        if (EPI == EPI_LSTM) {
            wrow[i] = (long long)(n >> 5) * P.H + tn * 32 + (n & 31);
            wok[i] = true;
        } else {
            wrow[i] = col0 + n;
            wok[i] = (col0 + n) < N;
        }
    }
    bool aok[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) aok[i] = (row0 + lr + 32 * i) < M;
    const bool a_col_ok = (row0 + akc) < M;   // k-major A: this thread's 4 output rows exist
    const bool b_col_ok = (col0 + bkc) < N;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int nchunks = 0;
    for (int s = 0; s < P.nseg; ++s) nchunks += (P.seg[s].K + BK - 1) / BK;

    // Per-thread source pointers of the current K-segment, advanced by one chunk per load (keeps the
    // 64-bit address arithmetic out of the loop); `fast` tiles (interior, K % 32 == 0) load unpredicated.
    // One register stage: chunk c+1 waits in it while chunk c is being contracted, is written to the
    // other LDS buffer in the middle of that contraction, and the stage is re-armed with chunk c+2 at
    // once - so a global load has a whole chunk of MFMAs (~4k cycles) to land.
    float4 ra[A_LD], rb[B_LD];
    const float *pa[A_LD], *pb[B_LD];
    long long stepA = BK, stepB = BK;
    int cs = 0, ck = 0, segK = 0;  // segment / k-offset of the chunk being loaded
    auto set_seg = [&](int si) __attribute__((always_inline)) {
        const DevSeg sg = P.seg[si];
        segK = sg.K;
        if constexpr (AKM) {
            stepA = (long long)BK * sg.lda;
#pragma unroll
            for (int i = 0; i < A_LD; ++i)
                pa[i] = sg.A + (long long)(akr + (1024 / BM) * i) * sg.lda + row0 + akc;
        } else {
#pragma unroll
            for (int i = 0; i < A_LD; ++i)
                pa[i] = sg.A + (long long)(row0 + lr + 32 * i) * sg.lda + lc;
        }
        if constexpr (BKM) {
            stepB = (long long)BK * sg.ldw;
#pragma unroll
            for (int i = 0; i < B_LD; ++i)
                pb[i] = sg.W + (long long)(bkr + (1024 / BN) * i) * sg.ldw + col0 + bkc;
        } else {
#pragma unroll
            for (int i = 0; i < B_LD; ++i) pb[i] = sg.W + wrow[i] * sg.ldw + lc;
        }
    };
    auto gload = [&](auto fastc) __attribute__((always_inline)) {
        constexpr bool FAST = decltype(fastc)::value;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            bool ok = true;
            if constexpr (!FAST) {
                if constexpr (AKM) ok = a_col_ok && (ck + akr + (1024 / BM) * i) < segK;
                else ok = aok[i] && (ck + lc) < segK;
            }
            ra[i] = ok ? *reinterpret_cast<const float4 *>(pa[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            pa[i] += stepA;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            bool ok = true;
            if constexpr (!FAST) {
                if constexpr (BKM) ok = b_col_ok && (ck + bkr + (1024 / BN) * i) < segK;
                else ok = wok[i] && (ck + lc) < segK;
            }
            rb[i] = ok ? *reinterpret_cast<const float4 *>(pb[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
            pb[i] += stepB;
        }
        ck += BK;
        if (ck >= segK) {
            ck = 0;
            if (++cs < P.nseg) set_seg(cs);
        }
    };
    auto sstore = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        float *a = As + BUF * TA::SIZE;
        if constexpr (AKM) {
#pragma unroll
            for (int i = 0; i < A_LD; ++i)
                *reinterpret_cast<float4 *>(a + (akr + (1024 / BM) * i) * (BM + 4) + akc) = ra[i];
        } else {
#pragma unroll
            for (int i = 0; i < A_LD; ++i)
                *reinterpret_cast<float4 *>(a + (lr + 32 * i) * LDT + lc) = ra[i];
        }
        float *b = Bs + BUF * TB::SIZE;
        if constexpr (BKM) {
#pragma unroll
            for (int i = 0; i < B_LD; ++i)
                *reinterpret_cast<float4 *>(b + (bkr + (1024 / BN) * i) * (BN + 4) + bkc) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < B_LD; ++i)
                *reinterpret_cast<float4 *>(b + (lr + 32 * i) * LDT + lc) = rb[i];
        }
    };

    // MFMA 32x32x2 operand maps: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].
    // Per 8-deep k-block a lane holds 4 consecutive k (k0 + 4*(l>>5) + t) of its row for both
    // operands; MFMA t consumes element t, so every k is covered exactly once.
    // Two fragment register sets: while the MFMAs of k-block kb run from one set, the fragments of
    // kb+1 (or of the next chunk's first k-block, after the barrier) are read into the other, so the
    // matrix pipe never waits for an LDS round trip - not even across a chunk boundary.
    static_assert(BK == 32, "the chunk body below is written for 4 k-blocks of 8");
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    float fa[2][4], fb[2][TN][4];
    auto lfrag = [&](auto bufc, auto kbc, auto setc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, kb = decltype(kbc)::value, ST = decltype(setc)::value;
        const float *at = As + BUF * TA::SIZE;
        const float *bt = Bs + BUF * TB::SIZE;
        if constexpr (AKM) {
#pragma unroll
            for (int e = 0; e < 4; ++e) fa[ST][e] = at[(kb * 8 + fk + e) * (BM + 4) + wm * 32 + frow];
        } else {
            const float4 v = *reinterpret_cast<const float4 *>(at + (wm * 32 + frow) * LDT + kb * 8 + fk);
            fa[ST][0] = v.x; fa[ST][1] = v.y; fa[ST][2] = v.z; fa[ST][3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if constexpr (BKM) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    fb[ST][j][e] = bt[(kb * 8 + fk + e) * (BN + 4) + (wn * TN + j) * 32 + frow];
            } else {
                const float4 v = *reinterpret_cast<const float4 *>(
                    bt + ((wn * TN + j) * 32 + frow) * LDT + kb * 8 + fk);
                fb[ST][j][0] = v.x; fb[ST][j][1] = v.y; fb[ST][j][2] = v.z; fb[ST][j][3] = v.w;
            }
        }
    };
    auto mma = [&](auto setc) __attribute__((always_inline)) {
        constexpr int ST = decltype(setc)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ST][e], fb[ST][j][e], acc[j], 0, 0, 0);
    };
    // split-K: this workgroup's chunk range [c_lo, c_lo + nchunks) of the problem's chunk sequence
    int c_lo = 0;
    if (ksplit > 1) {
        const int cps = (nchunks + ksplit - 1) / ksplit;
        c_lo = ks * cps;
        int mine = nchunks - c_lo;
        nchunks = mine < 0 ? 0 : (mine < cps ? mine : cps);
    }
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    // One chunk: LDS[CUR] holds it and fragment set 0 its first k-block; the register stage holds
    // chunk c+1 (if any).  Exactly one barrier per chunk, placed in front of the last k-block's MFMAs.
    auto chunk_body = [&](auto curc, auto nxtc, auto fastc, bool has1, bool has2) __attribute__((always_inline)) {
        lfrag(curc, I1{}, I1{});
        mma(I0{});
        if (has1) {
            sstore(nxtc);                 // LDS[NXT] was last read before the previous chunk's barrier
            if (has2) gload(fastc);
        }
        lfrag(curc, I2{}, I0{});
        mma(I1{});
        lfrag(curc, I3{}, I1{});
        mma(I0{});
        if (has1) {
            __syncthreads();              // every wave's stores of chunk c+1 have landed
            lfrag(nxtc, I0{}, I0{});
        }
        mma(I1{});
    };
    auto k_loop = [&](auto fastc) __attribute__((always_inline)) {
        if (nchunks == 0) return;
        {   // position the loader on chunk c_lo
            int skip = c_lo, si = 0;
            for (; si < P.nseg; ++si) {
                const int sc = (P.seg[si].K + BK - 1) / BK;
                if (skip < sc) break;
                skip -= sc;
            }
            set_seg(si);
            cs = si;
            ck = skip * BK;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) pa[i] += stepA * skip;
#pragma unroll
            for (int i = 0; i < B_LD; ++i) pb[i] += stepB * skip;
        }
        gload(fastc);                             // chunk 0
        sstore(I0{});
        if (nchunks > 1) gload(fastc);            // chunk 1 -> register stage
        __syncthreads();
        lfrag(I0{}, I0{}, I0{});
        for (int c = 0; c < nchunks; c += 2) {
            chunk_body(I0{}, I1{}, fastc, c + 1 < nchunks, c + 2 < nchunks);
            if (c + 1 < nchunks) chunk_body(I1{}, I0{}, fastc, c + 2 < nchunks, c + 3 < nchunks);
        }
        __syncthreads();                          // the epilogue re-uses the operand buffers
    };
    bool fast = (row0 + BM <= M) && (EPI == EPI_LSTM || col0 + BN <= N);
    for (int si = 0; si < P.nseg; ++si) fast = fast && (P.seg[si].K % BK) == 0;
    if (fast) k_loop(std::true_type{});
    else k_loop(std::false_type{});

    if constexpr (EPI == EPI_VOCAB) {
        epi_vocab_frag<TN, WN, BM>(P, acc, wm * 32, wn * TN * 32, wn, lane, row0, col0, tn, smem);
        if constexpr (WN > 1) {
            float *smx = smem;
            float *ssm = smem + WN * BM;
            int *six = reinterpret_cast<int *>(smem + 2 * WN * BM);
            __syncthreads();
            for (int row = tid; row < BM; row += 256) {
                const int gm = row0 + row;
                if (gm >= M) continue;
                float mx = smx[row];
                int ix = six[row];
#pragma unroll
                for (int w = 1; w < WN; ++w) {
                    const float ov = smx[w * BM + row];
                    const int oi = six[w * BM + row];
                    if (ov > mx || (ov == mx && oi < ix)) { mx = ov; ix = oi; }
                }
                float sm = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) {
                    const float wm_ = smx[w * BM + row];          // -inf: that wave had no valid column
                    if (wm_ > -INFINITY) sm += ssm[w * BM + row] * __expf(wm_ - mx);
                }
                const long long o = (long long)gm * P.ntile_total + tn;
                P.pmax[o] = mx;
                P.psum[o] = sm;
                P.pidx[o] = ix;
            }
        }
        return;
    }

    epi_stage_frag<TN>(Cs, LDC, acc, wm * 32, wn * TN * 32, lane);
    __syncthreads();
    if (EPI == EPI_LINEAR) epi_linear_tile<BM, BN>(P, Cs, row0, col0, tid, ks, ksplit);
    else if (EPI == EPI_LSTM) epi_lstm_tile<BM, BN>(P, Cs, row0, tn, tid);
}

// ---------------------------------------------------------------- XL tile: 256 x 128, one workgroup per CU
// For batch-sized NT problems.  4 waves stacked in M, each owning a 64 x 128 accumulator (2 x 4 MFMA
// fragments, 128 AGPRs): twice the MFMAs per staged byte of the 128 x 128 tile.  Operand chunks go
// (M0 carries the LDS destination of an LDS-DMA; it is written in the same asm statement that uses it, with the one
// wait state the ISA asks for between an SALU write of M0 and a memory instruction with the LDS modifier: hipcc's hazard
// recogniser does not look inside inline asm.)
// global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass) into a
// 3-deep ring (144 KB), so the DMA of chunk c+2 is in flight while chunk c is contracted and only a
// counted vmcnt (never 0 in steady state) precedes the one barrier per chunk.  The 12 DMA
// instructions of a chunk are issued one by one in the shadow of 8-MFMA groups.  An LDS-DMA writes
// lane-linear (wave-uniform base + 16 B * lane), which rules out padded rows: the tile image is
// [rows][8 slots of 16 B] with slot = k-quad ^ ((row >> 1) & 7), applied to the per-lane SOURCE address
// when staging and to the ds_read_b128 address when reading fragments (conflict-free for the
// instruction's four 16-lane groups).  The DMA is issued from inline asm (M0 = LDS address): the
// compiler would otherwise drain every DMA (vmcnt(0)) in front of the next ds_read.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_xl_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    constexpr int BM = 256, BN = 128, TN = 4;
    constexpr int TSA = BM * BK, TSB = BN * BK;         // floats per ring slot
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                     // [3][TSA]
    float *Bs = smem + 3 * TSA;           // [3][TSB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wm = tid >> 6;
    int pi, tm, tn, ks, ksplit;
    map_tile(L, pi, tm, tn, ks, ksplit);
    const DevProb &P = L.p[pi];
    const int M = P.M, N = P.N;
    const int row0 = tm * BM, col0 = tn * BN;

    // staging pieces (8 rows x 128 B each): A piece i of wave w = tile rows 64w + 8i .. +8, B piece i =
    // tile columns 32w + 8i .. +8; lane -> (row = lane>>3, slot = lane&7).  Rows past the edge re-read
    // the last valid row (their results are never stored).
    int arow[8];
    long long wrow[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = row0 + 64 * wm + 8 * i + (lane >> 3);
        arow[i] = r < M ? r : M - 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (EPI == EPI_LSTM) {
            wrow[i] = (long long)wm * P.H + tn * 32 + 8 * i + (lane >> 3);   // gate wm, unit tn*32 + ..
        } else {
            const int c = col0 + 32 * wm + 8 * i + (lane >> 3);
            wrow[i] = c < N ? c : N - 1;
        }
    }
    const int lq = lane & 7, lh = lane >> 4;   // (tile row >> 1) & 7 == (4*(i&1) + lh) for both operands
    const float *pa[8], *pb[4];
    int cs = 0, ck = 0, segK = 0;
    auto set_seg = [&](int si) __attribute__((always_inline)) {
        const DevSeg sg = P.seg[si];
        segK = sg.K;
#pragma unroll
        for (int i = 0; i < 8; ++i) pa[i] = sg.A + (long long)arow[i] * sg.lda + (lq ^ (4 * (i & 1) + lh)) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) pb[i] = sg.W + wrow[i] * sg.ldw + (lq ^ (4 * (i & 1) + lh)) * 4;
    };
    const unsigned a_lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)As + wm * 64 * BK * 4);
    const unsigned b_lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)Bs + wm * 32 * BK * 4);
    auto dma = [&](auto bufc, auto idxc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, IDX = decltype(idxc)::value;
        if constexpr (IDX < 8) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(a_lds0 + (BUF * TSA + 8 * IDX * BK) * 4), "v"(pa[IDX]) : "memory");
            pa[IDX] += BK;
        } else {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(b_lds0 + (BUF * TSB + 8 * (IDX - 8) * BK) * 4), "v"(pb[IDX - 8]) : "memory");
            pb[IDX - 8] += BK;
        }
        if constexpr (IDX == 11) {         // the chunk is fully issued: move the loader on
            ck += BK;
            if (ck >= segK) {
                ck = 0;
                if (++cs < P.nseg) set_seg(cs);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto dma_all = [&](auto bufc) __attribute__((always_inline)) {
        dma(bufc, I0{}); dma(bufc, I1{}); dma(bufc, I2{}); dma(bufc, I3{});
        dma(bufc, std::integral_constant<int, 4>{}); dma(bufc, std::integral_constant<int, 5>{});
        dma(bufc, std::integral_constant<int, 6>{}); dma(bufc, std::integral_constant<int, 7>{});
        dma(bufc, std::integral_constant<int, 8>{}); dma(bufc, std::integral_constant<int, 9>{});
        dma(bufc, std::integral_constant<int, 10>{}); dma(bufc, std::integral_constant<int, 11>{});
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    static_assert(BK == 32, "chunk body written for 4 k-blocks of 8");
    const int frow = lane & 31, khalf = lane >> 5;
    const int fsw = (frow >> 1) & 7;       // identical for rows frow + 32*j
    float4 fa[2][2], fb[2][TN];
    auto lfrag = [&](auto bufc, auto kbc, auto setc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, kb = decltype(kbc)::value, ST = decltype(setc)::value;
        const int slot = ((2 * kb + khalf) ^ fsw) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            fa[ST][i] = *reinterpret_cast<const float4 *>(As + BUF * TSA + (wm * 64 + i * 32 + frow) * BK + slot);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fb[ST][j] = *reinterpret_cast<const float4 *>(Bs + BUF * TSB + (j * 32 + frow) * BK + slot);
    };
    auto mma_e = [&](auto setc, auto ec) __attribute__((always_inline)) {
        constexpr int ST = decltype(setc)::value, e = decltype(ec)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float a[4] = {fa[ST][i].x, fa[ST][i].y, fa[ST][i].z, fa[ST][i].w};
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float b[4] = {fb[ST][j].x, fb[ST][j].y, fb[ST][j].z, fb[ST][j].w};
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[i][j], 0, 0, 0);
            }
        }
    };
    // one k-block (32 MFMAs) with up to 4 DMA pieces of chunk c+2 issued between its 8-MFMA groups
    auto mma_dma = [&](auto setc, auto bufc, auto basec, bool on) __attribute__((always_inline)) {
        constexpr int B0 = decltype(basec)::value;
        mma_e(setc, I0{}); if (on) dma(bufc, std::integral_constant<int, B0 + 0>{});
        mma_e(setc, I1{}); if (on) dma(bufc, std::integral_constant<int, B0 + 1>{});
        mma_e(setc, I2{}); if (on) dma(bufc, std::integral_constant<int, B0 + 2>{});
        mma_e(setc, I3{}); if (on) dma(bufc, std::integral_constant<int, B0 + 3>{});
    };
    int nchunks = 0;
    for (int s = 0; s < P.nseg; ++s) nchunks += P.seg[s].K / BK;
    // chunk c lives in ring slot c % 3.  Slot (c+2) % 3 was last read in chunk c-1, whose reads all
    // retired (lgkmcnt(0)) in front of that chunk's barrier - so its DMA may start right away.
    auto chunk_body = [&](auto curc, auto nxtc, auto nnc, bool has1, bool has2) __attribute__((always_inline)) {
        lfrag(curc, I1{}, I1{});
        mma_dma(I0{}, nnc, I0{}, has2);
        lfrag(curc, I2{}, I0{});
        mma_dma(I1{}, nnc, std::integral_constant<int, 4>{}, has2);
        lfrag(curc, I3{}, I1{});
        mma_dma(I0{}, nnc, std::integral_constant<int, 8>{}, has2);
        if (has1) {
            // chunk c+1 (issued one chunk ago) must have landed; chunk c+2's 12 pieces stay in flight
            if (has2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            lfrag(nxtc, I0{}, I0{});
        }
        mma_e(I1{}, I0{}); mma_e(I1{}, I1{}); mma_e(I1{}, I2{}); mma_e(I1{}, I3{});
    };
    set_seg(0);
    dma_all(I0{});
    if (nchunks > 1) {
        dma_all(I1{});
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    lfrag(I0{}, I0{}, I0{});
    for (int c = 0; c < nchunks; c += 3) {
        chunk_body(I0{}, I1{}, I2{}, c + 1 < nchunks, c + 2 < nchunks);
        if (c + 1 < nchunks) chunk_body(I1{}, I2{}, I0{}, c + 2 < nchunks, c + 3 < nchunks);
        if (c + 2 < nchunks) chunk_body(I2{}, I0{}, I1{}, c + 3 < nchunks, c + 4 < nchunks);
    }

    if constexpr (EPI == EPI_VOCAB) {
        epi_vocab_frag<TN, 1, BM>(P, acc[0], wm * 64, 0, 0, lane, row0, col0, tn, smem);
        epi_vocab_frag<TN, 1, BM>(P, acc[1], wm * 64 + 32, 0, 0, lane, row0, col0, tn, smem);
    } else if constexpr (EPI == EPI_LSTM) {
        epi_lstm_frag(P, acc[0], wm * 64, lane, row0, tn);
        epi_lstm_frag(P, acc[1], wm * 64 + 32, lane, row0, tn);
    } else {
        epi_linear_frag<4>(P, acc[0], wm * 64, 0, lane, row0, col0);
        epi_linear_frag<4>(P, acc[1], wm * 64 + 32, 0, lane, row0, col0);
    }
}

// ---------------------------------------------------------------- LD tile: 128 x 128 by LDS-DMA, two workgroups per CU
// The L tile's geometry (4 waves stacked in M, 32 x 128 accumulator each) fed like the XL tile: operands by
// global_load_lds_dwordx4 into two XOR-swizzled 32 KB buffers, no register stage and no ds_write pass (the
// register-staged loop spends ~9 % of its time moving the staged chunk VGPR -> LDS).  64 KB of LDS keep two
// workgroups on a CU, so an epilogue-heavy kernel - the vocabulary projection: 16 chunks of MFMAs, then per-row
// softmax statistics - still has a neighbour to hide its epilogue behind, which the XL tile does not.
// Chunk c+1's eight DMAs per wave are issued inside k-block 0 of chunk c (its buffer was last read before the
// previous barrier) and retired with vmcnt(0) in front of the one barrier per chunk.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_ld_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    constexpr int BM = 128, BN = 128, TN = 4;
    constexpr int TS = 128 * BK;                        // floats per operand buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                     // [2][TS]
    float *Bs = smem + 2 * TS;            // [2][TS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wm = tid >> 6;
    int pi, tm, tn, ks, ksplit;
    map_tile(L, pi, tm, tn, ks, ksplit);
    const DevProb &P = L.p[pi];
    const int M = P.M, N = P.N;
    const int row0 = tm * BM, col0 = tn * BN;

    // staging pieces (8 rows x 128 B): piece i of wave w = tile rows / columns 32w + 8i .. +8
    int arow[4];
    long long wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 32 * wm + 8 * i + (lane >> 3);
        arow[i] = r < M ? r : M - 1;
        if (EPI == EPI_LSTM) {
            wrow[i] = (long long)wm * P.H + tn * 32 + 8 * i + (lane >> 3);
        } else {
            const int c = col0 + 32 * wm + 8 * i + (lane >> 3);
            wrow[i] = c < N ? c : N - 1;
        }
    }
    const int lq = lane & 7, lh = lane >> 4;
    const float *pa[4], *pb[4];
    int cs = 0, ck = 0, segK = 0;
    auto set_seg = [&](int si) __attribute__((always_inline)) {
        const DevSeg sg = P.seg[si];
        segK = sg.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pa[i] = sg.A + (long long)arow[i] * sg.lda + (lq ^ (4 * (i & 1) + lh)) * 4;
            pb[i] = sg.W + wrow[i] * sg.ldw + (lq ^ (4 * (i & 1) + lh)) * 4;
        }
    };
    const unsigned a_lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)As + wm * 32 * BK * 4);
    const unsigned b_lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)Bs + wm * 32 * BK * 4);
    auto dma = [&](auto bufc, auto idxc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, IDX = decltype(idxc)::value;
        if constexpr (IDX < 4) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(a_lds0 + (BUF * TS + 8 * IDX * BK) * 4), "v"(pa[IDX]) : "memory");
            pa[IDX] += BK;
        } else {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(b_lds0 + (BUF * TS + 8 * (IDX - 4) * BK) * 4), "v"(pb[IDX - 4]) : "memory");
            pb[IDX - 4] += BK;
        }
        if constexpr (IDX == 7) {
            ck += BK;
            if (ck >= segK) {
                ck = 0;
                if (++cs < P.nseg) set_seg(cs);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;
    using I7 = std::integral_constant<int, 7>;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    static_assert(BK == 32, "chunk body written for 4 k-blocks of 8");
    const int frow = lane & 31, khalf = lane >> 5;
    const int fsw = (frow >> 1) & 7;
    float4 fa[2], fb[2][TN];
    auto lfrag = [&](auto bufc, auto kbc, auto setc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, kb = decltype(kbc)::value, ST = decltype(setc)::value;
        const int slot = ((2 * kb + khalf) ^ fsw) * 4;
        fa[ST] = *reinterpret_cast<const float4 *>(As + BUF * TS + (wm * 32 + frow) * BK + slot);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fb[ST][j] = *reinterpret_cast<const float4 *>(Bs + BUF * TS + (j * 32 + frow) * BK + slot);
    };
    auto mma_e = [&](auto setc, auto ec) __attribute__((always_inline)) {
        constexpr int ST = decltype(setc)::value, e = decltype(ec)::value;
        const float a[4] = {fa[ST].x, fa[ST].y, fa[ST].z, fa[ST].w};
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float b[4] = {fb[ST][j].x, fb[ST][j].y, fb[ST][j].z, fb[ST][j].w};
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[j], 0, 0, 0);
        }
    };
    auto mma = [&](auto setc) __attribute__((always_inline)) {
        mma_e(setc, I0{}); mma_e(setc, I1{}); mma_e(setc, I2{}); mma_e(setc, I3{});
    };
    int nchunks = 0;
    for (int s = 0; s < P.nseg; ++s) nchunks += P.seg[s].K / BK;
    auto chunk_body = [&](auto curc, auto nxtc, bool has1) __attribute__((always_inline)) {
        lfrag(curc, I1{}, I1{});
        // k-block 0: 4 MFMAs, then two DMA pieces of the next chunk, four times over
        mma_e(I0{}, I0{}); if (has1) { dma(nxtc, I0{}); dma(nxtc, I1{}); }
        mma_e(I0{}, I1{}); if (has1) { dma(nxtc, I2{}); dma(nxtc, I3{}); }
        mma_e(I0{}, I2{}); if (has1) { dma(nxtc, I4{}); dma(nxtc, I5{}); }
        mma_e(I0{}, I3{}); if (has1) { dma(nxtc, I6{}); dma(nxtc, I7{}); }
        lfrag(curc, I2{}, I0{});
        mma(I1{});
        lfrag(curc, I3{}, I1{});
        mma(I0{});
        if (has1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            lfrag(nxtc, I0{}, I0{});
        }
        mma(I1{});
    };
    set_seg(0);
    dma(I0{}, I0{}); dma(I0{}, I1{}); dma(I0{}, I2{}); dma(I0{}, I3{});
    dma(I0{}, I4{}); dma(I0{}, I5{}); dma(I0{}, I6{}); dma(I0{}, I7{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lfrag(I0{}, I0{}, I0{});
    for (int c = 0; c < nchunks; c += 2) {
        chunk_body(I0{}, I1{}, c + 1 < nchunks);
        if (c + 1 < nchunks) chunk_body(I1{}, I0{}, c + 2 < nchunks);
    }

    if constexpr (EPI == EPI_VOCAB) {
        epi_vocab_frag<TN, 1, BM>(P, acc, wm * 32, 0, 0, lane, row0, col0, tn, smem);
    } else if constexpr (EPI == EPI_LSTM) {
        epi_lstm_frag(P, acc, wm * 32, lane, row0, tn);
    } else {
        epi_linear_frag<4>(P, acc, wm * 32, 0, lane, row0, col0);
    }
}

// ---- epilogues for the C/D layout of v_mfma_f32_16x16x32_f16 (the split-f16 tile kernels, round 3) ----------------
// A wave's tile = 2 row blocks x NB column blocks of 16 x 16; acc[i][j] is an f32x4 whose element r on lane l is tile
// row frow0 + 16 i + 4 (l >> 4) + r, tile column fcol0 + 16 j + (l & 15): a lane holds 8 rows x NB columns, and a row's
// 16 NB columns sit on the 16 lanes of ONE DPP row - per-row reductions are four DPP steps, no swizzle.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Vocabulary statistics (max, arg-max, sum exp(x - max)) of the wave's 32 rows over its 16 NB columns (= the whole
// 128-column tile: the kernels that use this have one wave across N), straight from the registers.
template <int NB>
__device__ __forceinline__ void epi_vocab_frag16(const DevProb &P, f32x4 (&acc)[2][NB], int frow0, int lane, int row0,
                                                 int col0, int tn) {
    const int M = P.M, N = P.N;
    float bv[NB];
    bool cok[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int gn = col0 + j * 16 + (lane & 15);
        cok[j] = gn < N;
        bv[j] = cok[j] ? P.bias0[gn] : 0.f;
    }
    float mx[8], sm[8];
    int ix[8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = i * 4 + r;
            mx[e] = -INFINITY;
            ix[e] = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                acc[i][j][r] = cok[j] ? acc[i][j][r] + bv[j] : -INFINITY;      // acc now holds the logits
                const int gn = col0 + j * 16 + (lane & 15);
                if (acc[i][j][r] > mx[e]) { mx[e] = acc[i][j][r]; ix[e] = gn; }   // j ascending => smaller column wins ties
            }
        }
    if (P.C) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = row0 + frow0 + i * 16 + 4 * (lane >> 4) + r;
                if (gm < M) {
                    float *crow = P.C + (long long)gm * P.ld_logits + col0 + (lane & 15);
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        if (cok[j]) crow[j * 16] = acc[i][j][r];
                }
            }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) row16_argmax(mx[e], ix[e]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = i * 4 + r;
            sm[e] = 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j) sm[e] += cok[j] ? __expf(acc[i][j][r] - mx[e]) : 0.f;   // padded columns add 0
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm[e] = row16_sum(sm[e]);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int gm = row0 + frow0 + (e >> 2) * 16 + 4 * (lane >> 4) + (e & 3);
            if (gm < M) {
                const long long o = (long long)gm * P.ntile_total + tn;
                P.pmax[o] = mx[e];
                P.psum[o] = sm[e];
                P.pidx[o] = ix[e];
            }
        }
    }
}

// LSTM cells from a 32 x 128 gate-interleaved tile.  The kernels stage W so that tile column t = 32 gate + 16 hf + l
// holds (gate, unit 2 l + hf) of the tile's 32 units (h3_lstm_wrow): column block j = gate j >> 1, hf = j & 1, and lane l
// owns all four gates of the ADJACENT units 2 (l & 15), 2 (l & 15) + 1 for its 8 rows.  Every global operand of the
// epilogue (c_prev, the hoisted `pre` term, the embedding-table row, h / c out, the f16 planes of h) is then one
// float2 per lane and 16 lanes x 8 B = a full 128-byte segment per row - with one unit per lane (64-byte segments,
// twice the instructions) the att-LSTM's epilogue, which streams 16 KB per caption of `pre` and table rows, cost the
// kernel 14 us at B = 4096.  All loads of a row batch are issued before any arithmetic (see lstm_cells).
__device__ __forceinline__ int h3_lstm_wrow_in_gate(int t) { return 2 * (t & 15) + ((t >> 4) & 1); }

__device__ __forceinline__ void epi_lstm_frag16(const DevProb &P, f32x4 (&acc)[2][8], int frow0, int lane, int row0,
                                                int tn) {
    const int H = P.H, u0 = tn * 32 + 2 * (lane & 15);
    const bool has_b = P.bias0 != nullptr, has_pre = P.pre != nullptr, has_tab = P.tab != nullptr;
    float2 b[4];
    if (has_b) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float2 x = *reinterpret_cast<const float2 *>(P.bias0 + k * H + u0);
            const float2 y = *reinterpret_cast<const float2 *>(P.bias1 + k * H + u0);
            b[k] = make_float2(x.x + y.x, x.y + y.y);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {            // one row block = 4 rows per lane at a time (register budget)
        int gm[4];
        bool ok[4];
        long long tok[4];
        float2 cp[4], q[4][4], t[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gm[r] = row0 + frow0 + i * 16 + 4 * (lane >> 4) + r;
            ok[r] = gm[r] < P.M;
        }
        if (has_tab) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tok[r] = ok[r] ? P.tab_ids[(long long)gm[r] * P.tab_ids_stride] : 0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            cp[r] = ok[r] ? *reinterpret_cast<const float2 *>(P.c_prev + (long long)gm[r] * H + u0) : make_float2(0.f, 0.f);
        if (has_pre) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    q[r][k] = ok[r] ? *reinterpret_cast<const float2 *>(P.pre + (long long)gm[r] * 4 * H + k * H + u0)
                                    : make_float2(0.f, 0.f);
        }
        if (has_tab) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    t[r][k] = ok[r] ? *reinterpret_cast<const float2 *>(P.tab + tok[r] * 4 * H + k * H + u0)
                                    : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float h2[2], c2[2], ga[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                float g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    g[k] = acc[i][2 * k + hf][r];
                    // g += (b_ih + b_hh);  g += pre;  g += table row   (same order in every kernel: lstm_cells)
                    if (has_b) g[k] += hf ? b[k].y : b[k].x;
                    if (has_pre) g[k] += hf ? q[r][k].y : q[r][k].x;
                    if (has_tab) g[k] += hf ? t[r][k].y : t[r][k].x;
                }
                const float gi = isc_sigmoid(g[0]), gf = isc_sigmoid(g[1]), gg = isc_tanh(g[2]), go = isc_sigmoid(g[3]);
                c2[hf] = gf * (hf ? cp[r].y : cp[r].x) + gi * gg;
                h2[hf] = go * isc_tanh(c2[hf]);
                ga[hf][0] = gi; ga[hf][1] = gf; ga[hf][2] = gg; ga[hf][3] = go;
            }
            if (ok[r]) {
                const long long o = (long long)gm[r] * H + u0;
                *reinterpret_cast<float2 *>(P.c_out + o) = make_float2(c2[0], c2[1]);
                *reinterpret_cast<float2 *>(P.h_out + o) = make_float2(h2[0], h2[1]);
                if (P.h_hi) {
                    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                    const _Float16 a0 = (_Float16)h2[0], a1 = (_Float16)h2[1];
                    const long long po = plane_index(gm[r], u0, H);          // u0 even: the pair stays in its 32-block
                    h2v hi = {a0, a1};
                    h2v lo = {(_Float16)((h2[0] - (float)a0) * 2048.f), (_Float16)((h2[1] - (float)a1) * 2048.f)};
                    *reinterpret_cast<h2v *>(P.h_hi + po) = hi;
                    *reinterpret_cast<h2v *>(P.h_lo + po) = lo;
                }
                if (P.hmask) {
                    const uint8_t m0 = P.hmask[o], m1 = P.hmask[o + 1];
                    *reinterpret_cast<float2 *>(P.hdrop + o) =
                        make_float2(h2[0] * (float)m0 * P.mask_scale, h2[1] * (float)m1 * P.mask_scale);
                }
                if (P.gates_out) {
                    float *go_ = P.gates_out + (long long)gm[r] * 4 * H + u0;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        *reinterpret_cast<float2 *>(go_ + k * H) = make_float2(ga[0][k], ga[1][k]);
                }
            }
        }
    }
}

// Numerics status (isc_status): word 1 = a split-f16 linear launch that staged CALLER data as fp32 rows (the raw region
// features of the prologue, training-mode activations) produced a non-finite pre-activation - an operand at or beyond the
// f16 range (|x| >= 65520: its hi plane is inf), or a NaN / inf fed in.
ISC_STATUS_DECL(gemm)

// Linear epilogue (same feature set and template split as epi_linear_frag).  CHECK: flag non-finite pre-activations.
template <int NB, bool EDGE, bool FEAT, bool RELU, bool CHECK>
__device__ __forceinline__ void epi_linear_frag16_impl(const DevProb &P, f32x4 (&acc)[2][NB], int frow0, int fcol0,
                                                       int lane, int row0, int col0) {
    const int M = P.M, N = P.N;
    float b0[NB], b1[NB], b2[NB];
    bool cok[NB];
    const bool h0 = P.bias0 != nullptr, h1 = P.bias1 != nullptr, h2 = P.bias2 != nullptr;
    const bool f_acc = FEAT && P.accumulate, f_pre = FEAT && P.C_pre != nullptr, f_mask = FEAT && P.mask != nullptr;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int gn = col0 + fcol0 + j * 16 + (lane & 15);
        cok[j] = !EDGE || gn < N;
        b0[j] = (h0 && cok[j]) ? P.bias0[gn] : 0.f;
        b1[j] = (h1 && cok[j]) ? P.bias1[gn] : 0.f;
        b2[j] = (h2 && cok[j]) ? P.bias2[gn] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = row0 + frow0 + i * 16 + 4 * (lane >> 4) + r;
            if (EDGE && gm >= M) continue;
            float *crow = P.C + (long long)gm * P.ldc + col0 + fcol0 + (lane & 15);
            float prev[NB];
            if (f_acc) {                                   // all loads of the row before any arithmetic
#pragma unroll
                for (int j = 0; j < NB; ++j) prev[j] = (!EDGE || cok[j]) ? crow[j * 16] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (EDGE && !cok[j]) continue;
                float o = acc[i][j][r];
                if (h0) o += b0[j];
                if (h1) o += b1[j];
                if (h2) o += b2[j];
                if (f_acc) o += prev[j];
                if constexpr (CHECK) bad |= !(fabsf(o) <= 3.0e38f);
                if (RELU) o = isc_relu(o);
                if constexpr (FEAT) {
                    const int gn = col0 + fcol0 + j * 16 + (lane & 15);
                    if (f_pre) P.C_pre[(long long)gm * P.ldc + gn] = o;
                    if (f_mask) o = o * (float)P.mask[(long long)gm * N + gn] * P.mask_scale;
                }
                crow[j * 16] = o;
            }
        }
    if constexpr (CHECK) {
        if (__any(bad) && lane == 0) isc_flag_gemm(ISC_STATUS_WORD_LINEAR);
    }
}

template <int NB, bool CHECK>
__device__ __forceinline__ void epi_linear_frag16(const DevProb &P, f32x4 (&acc)[2][NB], int frow0, int fcol0, int lane,
                                                  int row0, int col0) {
    const bool feat = P.accumulate || P.C_pre || P.mask;
    const bool edge = !(row0 + frow0 + 32 <= P.M && col0 + fcol0 + 16 * NB <= P.N);
    const bool relu = P.relu != 0;
#define ISC_EPI_CASE(E, F, R) epi_linear_frag16_impl<NB, E, F, R, CHECK>(P, acc, frow0, fcol0, lane, row0, col0)
    if (edge) {
        if (relu) ISC_EPI_CASE(true, true, true); else ISC_EPI_CASE(true, true, false);
    } else if (feat) {
        if (relu) ISC_EPI_CASE(false, true, true); else ISC_EPI_CASE(false, true, false);
    } else {
        if (relu) ISC_EPI_CASE(false, false, true); else ISC_EPI_CASE(false, false, false);
    }
#undef ISC_EPI_CASE
}

// K-split form of the large split-f16 linear tile: workgroup (tile, ks) contracts its share of the k-blocks and stores the
// RAW partial tile to slab[ks] ([M, N] row-major); splitk_linear_kernel sums the slabs in fixed order and applies the
// epilogue.  For long contractions on few tiles - the classifier's dX over T x (B1 + B2) = 4160 rows is 132 tiles of
// K = 9984: a quarter of the chip's workgroup slots for 376 us; three slices of K fill it.
template <int NB, bool CHECK>
__device__ __forceinline__ void epi_slab_frag16(const DevProb &P, f32x4 (&acc)[2][NB], int frow0, int lane, int row0, int col0,
                                                int ks) {
    float *base = P.slab + (long long)ks * P.slab_stride;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = row0 + frow0 + i * 16 + 4 * (lane >> 4) + r;
            if (gm >= P.M) continue;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int gn = col0 + j * 16 + (lane & 15);
                if (gn >= P.N) continue;
                const float o = acc[i][j][r];
                if constexpr (CHECK) bad |= !(fabsf(o) <= 3.0e38f);
                base[(long long)gm * P.N + gn] = o;
            }
        }
    if constexpr (CHECK) {
        if (__any(bad) && lane == 0) isc_flag_gemm(ISC_STATUS_WORD_LINEAR);
    }
}

// ---------------------------------------------------------------- H3 tile: 128 x 128 on the f16 matrix cores
// fp32 operands, fp32 results, f16 MFMA rate.  Every operand value is split once into two f16 planes,
//     x = hi + lo * 2^-11,   hi = f16(x),   lo = f16((x - hi) * 2^11)
// which keeps >= 22 significant bits of x for |x| >= 2^-14 and an absolute error <= 2^-35 below (the subtraction is
// exact in fp32; tests/test_h3_math.py),
// and the contraction is taken as
//     C = sum hi_a hi_b  +  2^-11 * sum (hi_a lo_b + lo_a hi_b)
// on v_mfma_f32_32x32x16_f16: three MFMAs per 16-deep k-step instead of eight 32x32x2 fp32 ones at 1/16 the rate.
// f16 x f16 products are exact in the fp32 accumulator, the dropped lo_a lo_b term is <= 2^-22 relative, and the
// two sums live in separate accumulators so that the 2^-11 weight is applied once, in fp32, at the end.  Measured
// against an fp64 contraction the result is CLOSER than an fp32 fmaf chain (rms 1.0e-7 vs 2.6e-7 at K = 512,
// tools/h3_gemm_lab.hip) - the 1e-4 log-prob parity bound is not touched.  Domain: |x| < 65504 (f16 range of hi);
// beyond it hi is inf and the output NaN - loud, and far outside what the decoder's bounded activations reach.
// Geometry = the LD tile's (4 waves stacked in M, 32 x 128 accumulators, so the register epilogues are shared),
// operands by LDS-DMA from the pre-split planes: per 32-deep chunk four [128 x 32] f16 plane tiles of 8 KB
// (A hi, A lo, W hi, W lo), two buffers = 64 KB, two workgroups per CU.  64-byte LDS rows: slot s of row r holds
// k-octet q = s ^ ((r >> 2) & 3), which makes the ds_read_b128 fragment reads conflict-free.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// A-operand fragment from an fp32 image row (32 floats = 8 chunks of 16 B, chunk c at position c ^ sw): the 8 values of
// k-octet `oct` (chunks 2*oct, 2*oct+1) split into their hi / lo halfs in registers - what h3_split_kernel would have
// written, bit for bit.
__device__ __forceinline__ void h3_frag_from_f32(const char *row, int oct, int sw, h8 &hi, h8 &lo) {
    const float4 v0 = *reinterpret_cast<const float4 *>(row + (((2 * oct) ^ sw) * 16));
    const float4 v1 = *reinterpret_cast<const float4 *>(row + (((2 * oct + 1) ^ sw) * 16));
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)((x[e] - (float)h) * 2048.f);
    }
}

// Diagnostic build only (-DH3_STAMP=1, tools/h3_stamp.sh; the shipped library compiles none of it): s_memtime stamps
// around the main loop and epilogue of gemm_h3_kernel<vocab> (slot 0) and of gemm_h3x_kernel's LSTM form with K = 1024
// (slot 1: att-LSTM), K = 1536 (slot 2: lang-LSTM) and its linear form (slot 3), summed over waves into g_h3_stamp[slot] =
// {waves, setup, main loop, epilogue} (shader cycles), read back by isc_debug_h3_stamps (16 words).  The stamps go to
// memory nothing else reads.
#ifndef H3_STAMP
#define H3_STAMP 0
#endif
#if H3_STAMP
__device__ unsigned long long g_h3_stamp[16];
extern "C" int isc_debug_h3_stamps(unsigned long long *out16_host, int reset) {
    if (out16_host && hipMemcpyFromSymbol(out16_host, HIP_SYMBOL(g_h3_stamp), sizeof(g_h3_stamp)) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_h3_stamp), z, sizeof(z)) != hipSuccess) return 1;
    }
    return ISC_OK;
}
#endif

// AF32: some activation segment comes as fp32 rows (split after the fragment read); false = every segment has planes,
// and the per-buffer test is compiled out of the fragment loads (it cost the all-planes decode loop ~5-10 %).
template <int EPI, bool AF32>
__global__ __launch_bounds__(256, 2) void gemm_h3_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
#if H3_STAMP
    const unsigned long long stamp0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int BM = 128, BN = 128, TN = 4;
    constexpr int PA = 128 * 128, PB = 128 * 128;       // bytes per A / W image (row = 128 B: 32 hi | 32 lo halfs)
    constexpr int ST = PA + PB;                         // bytes per buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char *lds = reinterpret_cast<char *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wm = tid >> 6;
    int pi, tm, tn, ks, ksplit;
    map_tile(L, pi, tm, tn, ks, ksplit);
    const DevProb &P = L.p[pi];
    const int M = P.M, N = P.N, Kp = P.Kp;
    const int row0 = tm * BM, col0 = tn * BN;

    // staging in pieces of 8 rows x 128 B = whole lines of the interleaved plane layout (a row's 32 hi + 32 lo halfs
    // of a k-block): wave w issues A pieces 4w .. 4w+3 and W pieces 4w .. 4w+3; piece ii = image rows 8*ii .. +8,
    // 8 lanes per row.  (Pieces of 16 rows x 64 B - hi and lo fetched by different instructions - touched every line
    // twice and ran 9-14 % slower: tools/h3_gemm_lab.hip, r02.)  Image position p of row r holds chunk p ^ ((r>>1)&7).
    // W planes are K-packed [N, Kp]; A planes come by K-segment (P.ap: the split kernel's packed planes as one
    // segment, or the producers' per-tensor planes), switched at chunk boundaries.
    const _Float16 *src[8];
    int arow[4], aq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = 32 * wm + 8 * i + (lane >> 3);
        const int q = (lane & 7) ^ ((t >> 1) & 7);
        int ar = row0 + t;
        arow[i] = ar < M ? ar : M - 1;
        aq[i] = q * 8;
        long long wr;
        if (EPI == EPI_LSTM) {
            wr = (long long)(t >> 5) * P.H + tn * 32 + h3_lstm_wrow_in_gate(t);     // (gate, unit): see epi_lstm_frag16
        } else {
            const int c = col0 + t;
            wr = c < N ? c : N - 1;
        }
        src[4 + i] = P.Wh + wr * 2 * Kp + q * 8;
    }
    int cs = 0, ck = 0, segK = 0;
    bool seg_f32 = false;                // the segment has no planes: its fp32 rows are staged (128 B per row and chunk,
    unsigned f32_bufs = 0;               // like a plane row) and split after the fragment read; bit b: buffer b holds fp32
    auto set_aseg = [&](int si) __attribute__((always_inline)) {
        const DevASeg a = P.ap[si];
        segK = a.K;
        seg_f32 = AF32 && a.hi == nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            src[i] = seg_f32 ? reinterpret_cast<const _Float16 *>(P.seg[si].A + (long long)arow[i] * P.seg[si].lda) + aq[i]
                             : a.hi + (long long)arow[i] * a.ld + aq[i];
    };
    // this workgroup's share of the k-blocks (ksplit > 1, linear epilogue only: raw partial tiles go to slabs)
    const int nchunks_all = Kp / 32;
    const int c_lo = ksplit > 1 ? (int)((long long)nchunks_all * ks / ksplit) : 0;
    const int c_hi = ksplit > 1 ? (int)((long long)nchunks_all * (ks + 1) / ksplit) : nchunks_all;
    {
        int k0 = c_lo * 32;
        while (cs + 1 < P.nap && k0 >= P.ap[cs].K) { k0 -= P.ap[cs].K; ++cs; }
        set_aseg(cs);
        if (c_lo > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                src[i] += (long long)(k0 >> 5) * 64;
                src[4 + i] += (long long)c_lo * 64;
            }
            ck = k0;
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + wm * 4096);
    auto stage = [&](int buf) __attribute__((always_inline)) {
        f32_bufs = (f32_bufs & ~(1u << buf)) | ((seg_f32 ? 1u : 0u) << buf);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                             :: "s"(lds0 + buf * ST + p * PA + i * 1024), "v"(src[4 * p + i]) : "memory");
                src[4 * p + i] += 64;                     // next 32-k block: 32 hi + 32 lo halfs further
            }
        ck += 32;
        if (ck >= segK) {
            ck = 0;
            if (++cs < P.nap) set_aseg(cs);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // v_mfma_f32_16x16x32_f16 (round 3; the 32x32x16 form of rounds 1-2 ran 7-16 % slower on the same tile, staging and
    // LDS image - tools/h3_mfma16_lab.hip): the wave's 32 x 16 NB tile = 2 row blocks x NB column blocks, ONE MFMA per
    // block, product term and 32-k chunk.  Fragment of lane l: row (l & 15) of the block, k-octet (l >> 4) = 16-byte
    // chunk (l >> 4) of the row image's hi half, chunk 4 + (l >> 4) of its lo half (positions XOR-swizzled as staged).
    constexpr int NB = 2 * TN, NQ = NB / 2;             // NQ groups of two column blocks per chunk
    f32x4 acc0[2][NB], acc1[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    const int fr = lane & 15, fq = lane >> 4, fsw = (fr >> 1) & 7;
    const int ph = (fq ^ fsw) * 16, pl = ((4 + fq) ^ fsw) * 16;
    h8 ah[2][2], al[2][2], bh[2][2], bl[2][2];          // A: [set][row block]; W: [slot][column block of the group]
    auto lfragA = [&](int buf, auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        const char *base = lds + buf * ST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ra = (wm * 32 + i * 16 + fr) * 128;
            if (AF32 && ((f32_bufs >> buf) & 1u)) {
                h3_frag_from_f32(base + ra, fq, fsw, ah[S][i], al[S][i]);
            } else {
                ah[S][i] = *reinterpret_cast<const h8 *>(base + ra + ph);
                al[S][i] = *reinterpret_cast<const h8 *>(base + ra + pl);
            }
        }
    };
    auto lfragB = [&](int buf, auto grpc, auto slotc) __attribute__((always_inline)) {
        constexpr int Q = decltype(grpc)::value, SL = decltype(slotc)::value;
        const char *base = lds + buf * ST + PA;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rb = ((Q * 2 + j) * 16 + fr) * 128;
            bh[SL][j] = *reinterpret_cast<const h8 *>(base + rb + ph);
            bl[SL][j] = *reinterpret_cast<const h8 *>(base + rb + pl);
        }
    };
    auto mma = [&](auto setc, auto grpc, auto slotc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value, Q = decltype(grpc)::value, SL = decltype(slotc)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc0[i][Q * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[S][i], bh[SL][j], acc0[i][Q * 2 + j], 0, 0, 0);
                acc1[i][Q * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[S][i], bl[SL][j], acc1[i][Q * 2 + j], 0, 0, 0);
                acc1[i][Q * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[S][i], bh[SL][j], acc1[i][Q * 2 + j], 0, 0, 0);
            }
    };
    // one chunk's MFMAs with the W fragments of group q + 1 read under the MFMAs of group q; `between` runs in front of
    // the last group (the barrier, then the next chunk's A fragments into the other A set and its group-0 W fragments)
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto chunk_mma = [&](int cur, auto setc, auto between) __attribute__((always_inline)) {
        if constexpr (NQ == 4) {
            lfragB(cur, I1{}, I1{}); mma(setc, I0{}, I0{});
            lfragB(cur, I2{}, I0{}); mma(setc, I1{}, I1{});
            lfragB(cur, I3{}, I1{}); mma(setc, I2{}, I0{});
            between();
            mma(setc, I3{}, I1{});
        } else {
            static_assert(NQ == 2, "two or four groups of column blocks per chunk");
            lfragB(cur, I1{}, I1{}); mma(setc, I0{}, I0{});
            between();
            mma(setc, I1{}, I1{});
        }
    };
    const int nchunks = c_hi - c_lo;
#if H3_STAMP
    const unsigned long long stamp1 = __builtin_amdgcn_s_memtime();
#endif
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lfragA(0, I0{});
    lfragB(0, I0{}, I0{});
    // per chunk: the second half's W fragments are read under the first half's MFMAs, the next chunk's A fragments
    // (into the other A set) and first-half W fragments under the second half's
    auto chunk = [&](int c, auto setc, auto nsetc) __attribute__((always_inline)) {
        const int cur = c & 1, nxt = cur ^ 1;
        const bool has1 = c + 1 < nchunks;
        if (has1) stage(nxt);                 // buffer nxt was last read in front of the previous barrier
        chunk_mma(cur, setc, [&]() __attribute__((always_inline)) {
            if (has1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                lfragA(nxt, nsetc);
                lfragB(nxt, I0{}, I0{});
            }
        });
    };
    for (int c = 0; c < nchunks; c += 2) {
        chunk(c, I0{}, I1{});
        if (c + 1 < nchunks) chunk(c + 1, I1{}, I0{});
    }
#if H3_STAMP
    const unsigned long long stamp2 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc0[i][j][r] = fmaf(acc1[i][j][r], 1.f / 2048.f, acc0[i][j][r]);

    if constexpr (EPI == EPI_VOCAB) {
        epi_vocab_frag16<NB>(P, acc0, wm * 32, lane, row0, col0, tn);
    } else if constexpr (EPI == EPI_LSTM) {
        epi_lstm_frag16(P, acc0, wm * 32, lane, row0, tn);
    } else {
        if (ksplit > 1) epi_slab_frag16<NB, AF32>(P, acc0, wm * 32, lane, row0, col0, ks);
        else epi_linear_frag16<NB, AF32>(P, acc0, wm * 32, 0, lane, row0, col0);
    }
#if H3_STAMP
    if (EPI == EPI_VOCAB) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the epilogue's stores have left
        const unsigned long long stamp3 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            atomicAdd(&g_h3_stamp[0], 1ull);
            atomicAdd(&g_h3_stamp[1], stamp1 - stamp0);
            atomicAdd(&g_h3_stamp[2], stamp2 - stamp1);
            atomicAdd(&g_h3_stamp[3], stamp3 - stamp2);
        }
    }
#endif
}

// H3 on a 256 x 128 tile with EIGHT waves (32 x 128 accumulators, the same fragment code) and three 48 KB buffers:
// one workgroup per CU keeps two chunks of DMA in flight (96 KB against the 64 KB of two 128-row workgroups with one
// chunk each) - these launches are bound by the DMA round trip, not by the matrix pipe: [4096 x 9984 x 512] 178 -> 137 us,
// [4096 x 2048 x 1536] 97 -> 76 us (tools/h3_gemm_lab.hip).  Used when the launch has >= 224 such tiles.
template <int EPI, bool AF32>
__global__ __launch_bounds__(512) void gemm_h3x_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
#if H3_STAMP
    const unsigned long long stamp0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int BM = 256, BN = 128, TN = 4;
    constexpr int PA = 256 * 128, PB = 128 * 128;       // bytes per A / W image (row = 128 B: 32 hi | 32 lo halfs)
    constexpr int ST = PA + PB;                         // bytes per buffer (48 KB)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char *lds = reinterpret_cast<char *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wm = tid >> 6;
    int pi, tm, tn, ks, ksplit;
    map_tile(L, pi, tm, tn, ks, ksplit);
    const DevProb &P = L.p[pi];
    const int M = P.M, N = P.N, Kp = P.Kp;
    const int row0 = tm * BM, col0 = tn * BN;

    // staging pieces of 8 rows x 128 B (whole lines, see gemm_h3_kernel): A image 32 pieces (four per wave), W image 16
    // (two per wave).  A by K-segment.
    const _Float16 *src[6];
    int arow[4], aq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = 32 * wm + 8 * i + (lane >> 3);
        int ar = row0 + t;
        arow[i] = ar < M ? ar : M - 1;
        aq[i] = ((lane & 7) ^ ((t >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = 16 * wm + 8 * i + (lane >> 3);
        const int q = (lane & 7) ^ ((t >> 1) & 7);
        long long wr;
        if (EPI == EPI_LSTM) {
            wr = (long long)(t >> 5) * P.H + tn * 32 + h3_lstm_wrow_in_gate(t);     // (gate, unit): see epi_lstm_frag16
        } else {
            const int c = col0 + t;
            wr = c < N ? c : N - 1;
        }
        src[4 + i] = P.Wh + wr * 2 * Kp + q * 8;
    }
    int cs = 0, ck = 0, segK = 0;
    bool seg_f32 = false;                // the segment has no planes: its fp32 rows are staged (128 B per row and chunk,
    unsigned f32_bufs = 0;               // like a plane row) and split after the fragment read; bit b: buffer b holds fp32
    auto set_aseg = [&](int si) __attribute__((always_inline)) {
        const DevASeg a = P.ap[si];
        segK = a.K;
        seg_f32 = AF32 && a.hi == nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            src[i] = seg_f32 ? reinterpret_cast<const _Float16 *>(P.seg[si].A + (long long)arow[i] * P.seg[si].lda) + aq[i]
                             : a.hi + (long long)arow[i] * a.ld + aq[i];
    };
    set_aseg(0);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
    const unsigned wv = __builtin_amdgcn_readfirstlane((unsigned)wm);
    auto dma1 = [&](unsigned dst, const _Float16 *&p) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(p) : "memory");
        p += 64;
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        const unsigned b = lds0 + buf * ST;
        f32_bufs = (f32_bufs & ~(1u << buf)) | ((seg_f32 ? 1u : 0u) << buf);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma1(b + (4 * wv + i) * 1024, src[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma1(b + PA + (2 * wv + i) * 1024, src[4 + i]);
        ck += 32;
        if (ck >= segK) {
            ck = 0;
            if (++cs < P.nap) set_aseg(cs);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // v_mfma_f32_16x16x32_f16 (round 3; the 32x32x16 form of rounds 1-2 ran 7-16 % slower on the same tile, staging and
    // LDS image - tools/h3_mfma16_lab.hip): the wave's 32 x 16 NB tile = 2 row blocks x NB column blocks, ONE MFMA per
    // block, product term and 32-k chunk.  Fragment of lane l: row (l & 15) of the block, k-octet (l >> 4) = 16-byte
    // chunk (l >> 4) of the row image's hi half, chunk 4 + (l >> 4) of its lo half (positions XOR-swizzled as staged).
    constexpr int NB = 2 * TN, NQ = NB / 2;             // NQ groups of two column blocks per chunk
    f32x4 acc0[2][NB], acc1[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    const int fr = lane & 15, fq = lane >> 4, fsw = (fr >> 1) & 7;
    const int ph = (fq ^ fsw) * 16, pl = ((4 + fq) ^ fsw) * 16;
    h8 ah[2][2], al[2][2], bh[2][2], bl[2][2];          // A: [set][row block]; W: [slot][column block of the group]
    auto lfragA = [&](int buf, auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        const char *base = lds + buf * ST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ra = (wm * 32 + i * 16 + fr) * 128;
            if (AF32 && ((f32_bufs >> buf) & 1u)) {
                h3_frag_from_f32(base + ra, fq, fsw, ah[S][i], al[S][i]);
            } else {
                ah[S][i] = *reinterpret_cast<const h8 *>(base + ra + ph);
                al[S][i] = *reinterpret_cast<const h8 *>(base + ra + pl);
            }
        }
    };
    auto lfragB = [&](int buf, auto grpc, auto slotc) __attribute__((always_inline)) {
        constexpr int Q = decltype(grpc)::value, SL = decltype(slotc)::value;
        const char *base = lds + buf * ST + PA;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rb = ((Q * 2 + j) * 16 + fr) * 128;
            bh[SL][j] = *reinterpret_cast<const h8 *>(base + rb + ph);
            bl[SL][j] = *reinterpret_cast<const h8 *>(base + rb + pl);
        }
    };
    auto mma = [&](auto setc, auto grpc, auto slotc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value, Q = decltype(grpc)::value, SL = decltype(slotc)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc0[i][Q * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[S][i], bh[SL][j], acc0[i][Q * 2 + j], 0, 0, 0);
                acc1[i][Q * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[S][i], bl[SL][j], acc1[i][Q * 2 + j], 0, 0, 0);
                acc1[i][Q * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[S][i], bh[SL][j], acc1[i][Q * 2 + j], 0, 0, 0);
            }
    };
    // one chunk's MFMAs with the W fragments of group q + 1 read under the MFMAs of group q; `between` runs in front of
    // the last group (the barrier, then the next chunk's A fragments into the other A set and its group-0 W fragments)
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto chunk_mma = [&](int cur, auto setc, auto between) __attribute__((always_inline)) {
        if constexpr (NQ == 4) {
            lfragB(cur, I1{}, I1{}); mma(setc, I0{}, I0{});
            lfragB(cur, I2{}, I0{}); mma(setc, I1{}, I1{});
            lfragB(cur, I3{}, I1{}); mma(setc, I2{}, I0{});
            between();
            mma(setc, I3{}, I1{});
        } else {
            static_assert(NQ == 2, "two or four groups of column blocks per chunk");
            lfragB(cur, I1{}, I1{}); mma(setc, I0{}, I0{});
            between();
            mma(setc, I1{}, I1{});
        }
    };
    // three buffers, two chunks in flight (6 DMAs per wave and chunk): vmcnt(6) leaves the younger chunk outstanding
    const int nchunks = Kp / 32;
#if H3_STAMP
    const unsigned long long stamp1 = __builtin_amdgcn_s_memtime();
#endif
    stage(0);
    if (nchunks > 1) stage(1);
    if (nchunks > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lfragA(0, I0{});
    lfragB(0, I0{}, I0{});
    int cur = 0;
    auto chunk = [&](int c, auto setc, auto nsetc) __attribute__((always_inline)) {
        const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
        if (c + 2 < nchunks) stage(nn);           // buffer nn was last read in front of the previous barrier
        chunk_mma(cur, setc, [&]() __attribute__((always_inline)) {
            if (c + 1 < nchunks) {
                if (c + 2 < nchunks) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                lfragA(nxt, nsetc);
                lfragB(nxt, I0{}, I0{});
            }
        });
        cur = nxt;
    };
    for (int c = 0; c < nchunks; c += 2) {
        chunk(c, I0{}, I1{});
        if (c + 1 < nchunks) chunk(c + 1, I1{}, I0{});
    }
#if H3_STAMP
    const unsigned long long stamp2 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc0[i][j][r] = fmaf(acc1[i][j][r], 1.f / 2048.f, acc0[i][j][r]);

    if constexpr (EPI == EPI_VOCAB) {
        epi_vocab_frag16<NB>(P, acc0, wm * 32, lane, row0, col0, tn);
    } else if constexpr (EPI == EPI_LSTM) {
        epi_lstm_frag16(P, acc0, wm * 32, lane, row0, tn);
    } else {
        epi_linear_frag16<NB, AF32>(P, acc0, wm * 32, 0, lane, row0, col0);
    }
#if H3_STAMP
    if (EPI != EPI_VOCAB) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the epilogue's stores have left
        const unsigned long long stamp3 = __builtin_amdgcn_s_memtime();
        const int slot = EPI == EPI_LSTM ? (Kp == 1024 ? 1 : Kp == 1536 ? 2 : -1) : 3;
        if (lane == 0 && slot >= 0) {
            atomicAdd(&g_h3_stamp[4 * slot + 0], 1ull);
            atomicAdd(&g_h3_stamp[4 * slot + 1], stamp1 - stamp0);
            atomicAdd(&g_h3_stamp[4 * slot + 2], stamp2 - stamp1);
            atomicAdd(&g_h3_stamp[4 * slot + 3], stamp3 - stamp2);
        }
    }
#endif
}

#define H3M_NBUF 4
// H3 on the 64 x 128 geometry (2 x 2 waves, 32 x 64 accumulators each; linear epilogue): launches whose 128-row
// tiling would leave CUs idle - the per-step projections with N = 512 at 4096 rows are 256 tiles of 64 x 128.
// Per chunk: A planes 2 x 4 KB, W planes 2 x 8 KB; four buffers = 96 KB, one workgroup per CU.
template <bool AF32>
__global__ __launch_bounds__(256) void gemm_h3m_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    rows_kernarg_warm<ROWS_KERNARG_LINES(DevLaunch)>();      // (one batch of scalar loads for the launch descriptor: common.h)
    constexpr int TN = 2;
    constexpr int PA = 64 * 128, PB = 128 * 128;        // bytes per A / W image (row = 128 B: 32 hi | 32 lo halfs)
    constexpr int ST = PA + PB;                         // bytes per buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char *lds = reinterpret_cast<char *>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6, wm = w >> 1, wn = w & 1;
    int pi, tm, tn, ks, ksplit;
    map_tile(L, pi, tm, tn, ks, ksplit);
    const DevProb &P = L.p[pi];
    const int M = P.M, N = P.N, Kp = P.Kp;
    const int row0 = tm * 64, col0 = tn * 128;

    // staging pieces of 8 rows x 128 B (whole lines, see gemm_h3_kernel): A image two pieces per wave (by K-segment),
    // W image four
    const _Float16 *src[6];
    int arow[2], aq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = 16 * w + 8 * i + (lane >> 3);
        int ar = row0 + t;
        arow[i] = ar < M ? ar : M - 1;
        aq[i] = ((lane & 7) ^ ((t >> 1) & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = 32 * w + 8 * i + (lane >> 3);
        const int q = (lane & 7) ^ ((t >> 1) & 7);
        int c = col0 + t;
        c = c < N ? c : N - 1;
        src[2 + i] = P.Wh + (long long)c * 2 * Kp + q * 8;
    }
    int cs = 0, ck = 0, segK = 0;
    bool seg_f32 = false;                // see gemm_h3_kernel
    unsigned f32_bufs = 0;
    auto set_aseg = [&](int si) __attribute__((always_inline)) {
        const DevASeg a = P.ap[si];
        segK = a.K;
        seg_f32 = AF32 && a.hi == nullptr;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            src[i] = seg_f32 ? reinterpret_cast<const _Float16 *>(P.seg[si].A + (long long)arow[i] * P.seg[si].lda) + aq[i]
                             : a.hi + (long long)arow[i] * a.ld + aq[i];
    };
    set_aseg(0);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
    const unsigned wv = __builtin_amdgcn_readfirstlane((unsigned)w);
    auto dma1 = [&](unsigned dst, const _Float16 *&p) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(p) : "memory");
        p += 64;
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        const unsigned b = lds0 + buf * ST;
        f32_bufs = (f32_bufs & ~(1u << buf)) | ((seg_f32 ? 1u : 0u) << buf);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma1(b + (2 * wv + i) * 1024, src[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma1(b + PA + (4 * wv + i) * 1024, src[2 + i]);
        ck += 32;
        if (ck >= segK) {
            ck = 0;
            if (++cs < P.nap) set_aseg(cs);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // v_mfma_f32_16x16x32_f16 (round 3; the 32x32x16 form of rounds 1-2 ran 7-16 % slower on the same tile, staging and
    // LDS image - tools/h3_mfma16_lab.hip): the wave's 32 x 16 NB tile = 2 row blocks x NB column blocks, ONE MFMA per
    // block, product term and 32-k chunk.  Fragment of lane l: row (l & 15) of the block, k-octet (l >> 4) = 16-byte
    // chunk (l >> 4) of the row image's hi half, chunk 4 + (l >> 4) of its lo half (positions XOR-swizzled as staged).
    constexpr int NB = 2 * TN, NQ = NB / 2;             // NQ groups of two column blocks per chunk
    f32x4 acc0[2][NB], acc1[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    const int fr = lane & 15, fq = lane >> 4, fsw = (fr >> 1) & 7;
    const int ph = (fq ^ fsw) * 16, pl = ((4 + fq) ^ fsw) * 16;
    h8 ah[2][2], al[2][2], bh[2][2], bl[2][2];          // A: [set][row block]; W: [slot][column block of the group]
    auto lfragA = [&](int buf, auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
        const char *base = lds + buf * ST;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ra = (wm * 32 + i * 16 + fr) * 128;
            if (AF32 && ((f32_bufs >> buf) & 1u)) {
                h3_frag_from_f32(base + ra, fq, fsw, ah[S][i], al[S][i]);
            } else {
                ah[S][i] = *reinterpret_cast<const h8 *>(base + ra + ph);
                al[S][i] = *reinterpret_cast<const h8 *>(base + ra + pl);
            }
        }
    };
    auto lfragB = [&](int buf, auto grpc, auto slotc) __attribute__((always_inline)) {
        constexpr int Q = decltype(grpc)::value, SL = decltype(slotc)::value;
        const char *base = lds + buf * ST + PA;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rb = (wn * 64 + (Q * 2 + j) * 16 + fr) * 128;
            bh[SL][j] = *reinterpret_cast<const h8 *>(base + rb + ph);
            bl[SL][j] = *reinterpret_cast<const h8 *>(base + rb + pl);
        }
    };
    auto mma = [&](auto setc, auto grpc, auto slotc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value, Q = decltype(grpc)::value, SL = decltype(slotc)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc0[i][Q * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[S][i], bh[SL][j], acc0[i][Q * 2 + j], 0, 0, 0);
                acc1[i][Q * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[S][i], bl[SL][j], acc1[i][Q * 2 + j], 0, 0, 0);
                acc1[i][Q * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[S][i], bh[SL][j], acc1[i][Q * 2 + j], 0, 0, 0);
            }
    };
    // one chunk's MFMAs with the W fragments of group q + 1 read under the MFMAs of group q; `between` runs in front of
    // the last group (the barrier, then the next chunk's A fragments into the other A set and its group-0 W fragments)
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    auto chunk_mma = [&](int cur, auto setc, auto between) __attribute__((always_inline)) {
        if constexpr (NQ == 4) {
            lfragB(cur, I1{}, I1{}); mma(setc, I0{}, I0{});
            lfragB(cur, I2{}, I0{}); mma(setc, I1{}, I1{});
            lfragB(cur, I3{}, I1{}); mma(setc, I2{}, I0{});
            between();
            mma(setc, I3{}, I1{});
        } else {
            static_assert(NQ == 2, "two or four groups of column blocks per chunk");
            lfragB(cur, I1{}, I1{}); mma(setc, I0{}, I0{});
            between();
            mma(setc, I1{}, I1{});
        }
    };
    // H3M_NBUF buffers, H3M_NBUF - 1 chunks in flight: these launches run one workgroup per CU, so the DMA round trip
    // has to be covered by the workgroup's own prefetch depth, not by a neighbour.  (Round 3: six buffers = five chunks
    // in flight measured the same as four / three - gate sum [4096 x 512 x 1024] 31.9 us either way - so the round trip
    // is covered; what these N = 512 launches pay is the ISSUE of their LDS-DMA pieces: six per wave and chunk against
    // 24 MFMAs, where the 256-row kernel has 48.)
    const int nchunks = Kp / 32;
    constexpr int NBUF = H3M_NBUF, AHEAD = NBUF - 1;
    auto wait_for = [&](int in_flight) __attribute__((always_inline)) {   // stages that may still be outstanding (6 DMAs each)
        if (in_flight >= 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (in_flight == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (in_flight == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (in_flight == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert(AHEAD <= 5 && AHEAD >= 2, "wait_for covers up to 4 younger stages");
#pragma unroll
    for (int q = 0; q < AHEAD; ++q)
        if (q < nchunks) stage(q);
    wait_for((nchunks < AHEAD ? nchunks : AHEAD) - 1);
    __syncthreads();
    lfragA(0, I0{});
    lfragB(0, I0{}, I0{});
    int cur = 0;                                       // ring position of chunk c
    auto chunk = [&](int c, auto setc, auto nsetc) __attribute__((always_inline)) {
        const int nxt = cur + 1 == NBUF ? 0 : cur + 1;
        int far = cur + AHEAD;                         // where chunk c + AHEAD goes: the buffer chunk c - 1 left,
        if (far >= NBUF) far -= NBUF;                  // last read in front of the previous barrier
        if (c + AHEAD < nchunks) stage(far);
        chunk_mma(cur, setc, [&]() __attribute__((always_inline)) {
            if (c + 1 < nchunks) {
                const int last = nchunks - 1 < c + AHEAD ? nchunks - 1 : c + AHEAD;
                wait_for(last - (c + 1));
                __syncthreads();
                lfragA(nxt, nsetc);
                lfragB(nxt, I0{}, I0{});
            }
        });
        cur = nxt;
    };
    for (int c = 0; c < nchunks; c += 2) {
        chunk(c, I0{}, I1{});
        if (c + 1 < nchunks) chunk(c + 1, I1{}, I0{});
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc0[i][j][r] = fmaf(acc1[i][j][r], 1.f / 2048.f, acc0[i][j][r]);
    epi_linear_frag16<NB, AF32>(P, acc0, wm * 32, wn * 64, lane, row0, col0);
}

// Operand split in front of gemm_h3_kernel: gathers the K-segments of one operand into the two packed f16 planes
// [rows, Kp].  One thread per 8 consecutive k (two float4 loads, one 16-byte store per plane).
struct SplitJob {
    const float *src[ISC_MAX_SEG];
    int ld[ISC_MAX_SEG];
    int kstart[ISC_MAX_SEG];
    int nseg, rows, Kp, first_block;
    _Float16 *hi, *lo;
    int transposed, _pad;             // source segments are [K_s, rows] (k-major): planes of the TRANSPOSE are built
};
#define H3_MAX_JOBS 12
struct SplitLaunch {
    SplitJob j[H3_MAX_JOBS];
    int njobs;
};

__global__ __launch_bounds__(256) void h3_split_kernel(const SplitLaunch S) {
    int ji = 0;
#pragma unroll
    for (int i = 1; i < H3_MAX_JOBS; ++i)
        if (i < S.njobs && (int)blockIdx.x >= S.j[i].first_block) ji = i;
    const SplitJob &J = S.j[ji];
    const int k8n = J.Kp >> 3;
    const long long idx = (long long)(blockIdx.x - J.first_block) * 256 + threadIdx.x;
    if (J.transposed) {
        // source segments are [K_s, rows] (k-major: the backward pass's W for dX = dY W, and both operands of dW = dY^T X);
        // plane row n = column n of the source.  One workgroup transposes a tile of 32 k x 64 n through LDS: the reads are
        // 256-byte row pieces, and each output row receives its whole 128-byte line (32 hi + 32 lo halfs of the k-block).
        __shared__ float t[32][65];
        const int ntile = (J.rows + 63) >> 6;
        const int bid = blockIdx.x - J.first_block;
        const int kb = bid / ntile, n0 = (bid - kb * ntile) * 64, kq = kb * 32;
        const float *sp = J.src[0];
        int ld = J.ld[0], k0 = 0;
#pragma unroll
        for (int sg = 1; sg < ISC_MAX_SEG; ++sg)
            if (sg < J.nseg && kq >= J.kstart[sg]) { sp = J.src[sg]; ld = J.ld[sg]; k0 = J.kstart[sg]; }
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k = ty * 8 + r;
            t[k][tx] = n0 + tx < J.rows ? sp[(long long)(kq - k0 + k) * ld + n0 + tx] : 0.f;
        }
        __syncthreads();
        // thread -> (n = tid / 4, 8 consecutive k = 8 * (tid % 4)): four threads write one row's 64 B of hi and 64 B of lo
        const int n = threadIdx.x >> 2, q = (threadIdx.x & 3) * 8;
        if (n0 + n < J.rows) {
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = t[q + e][n];
                const _Float16 h = (_Float16)x;
                hi[e] = h;
                lo[e] = (_Float16)((x - (float)h) * 2048.f);
            }
            const long long o = plane_index(n0 + n, kq + q, J.Kp);
            *reinterpret_cast<h8 *>(J.hi + o) = hi;
            *reinterpret_cast<h8 *>(J.lo + o) = lo;
        }
        return;
    }
    if (idx >= (long long)J.rows * k8n) return;
    const int row = (int)(idx / k8n);
    int k = (int)(idx - (long long)row * k8n) * 8;
    const long long o = plane_index(row, k, J.Kp);        // 8 consecutive k never leave their 32-block
    const float *sp = J.src[0];
    int ld = J.ld[0], k0 = 0;
#pragma unroll
    for (int s = 1; s < ISC_MAX_SEG; ++s)
        if (s < J.nseg && k >= J.kstart[s]) { sp = J.src[s]; ld = J.ld[s]; k0 = J.kstart[s]; }
    const float *p = sp + (long long)row * ld + (k - k0);
    const float4 v0 = *reinterpret_cast<const float4 *>(p), v1 = *reinterpret_cast<const float4 *>(p + 4);
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)((x[e] - (float)h) * 2048.f);
    }
    *reinterpret_cast<h8 *>(J.hi + o) = hi;
    *reinterpret_cast<h8 *>(J.lo + o) = lo;
}

// ---------------------------------------------------------------- MD tile: 64 x 128 by LDS-DMA (linear epilogue)
// The LD scheme on the M geometry (2 x 2 waves, 32 x 64 accumulator each) for the per-step projections with
// N = 512: their 64-row tiles are what balances 4096 x 512 outputs over 256 CUs, and at one to three workgroups
// per CU the register-staged form of that tile leaves the matrix pipe idle over its chunk-boundary round trips.
__global__ __launch_bounds__(256) void gemm_md_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    constexpr int TN = 2;
    constexpr int TSA = 64 * BK, TSB = 128 * BK;      // floats per operand buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                     // [2][TSA]
    float *Bs = smem + 2 * TSA;           // [2][TSB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = tid >> 6, wm = w >> 1, wn = w & 1;
    int pi, tm, tn, ks, ksplit;
    map_tile(L, pi, tm, tn, ks, ksplit);
    const DevProb &P = L.p[pi];
    const int M = P.M, N = P.N;
    const int row0 = tm * 64, col0 = tn * 128;

    // staging pieces (8 rows x 128 B): A pieces 2w, 2w+1 = tile rows 16w + 8i; B pieces 4w .. 4w+3 = columns 32w + 8i
    int arow[2];
    long long wrow[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = row0 + 16 * w + 8 * i + (lane >> 3);
        arow[i] = r < M ? r : M - 1;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = col0 + 32 * w + 8 * i + (lane >> 3);
        wrow[i] = c < N ? c : N - 1;
    }
    const int lq = lane & 7, lh = lane >> 4;
    const float *pa[2], *pb[4];
    int cs = 0, ck = 0, segK = 0;
    auto set_seg = [&](int si) __attribute__((always_inline)) {
        const DevSeg sg = P.seg[si];
        segK = sg.K;
#pragma unroll
        for (int i = 0; i < 2; ++i) pa[i] = sg.A + (long long)arow[i] * sg.lda + (lq ^ (4 * (i & 1) + lh)) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) pb[i] = sg.W + wrow[i] * sg.ldw + (lq ^ (4 * (i & 1) + lh)) * 4;
    };
    const unsigned a_lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)As + w * 16 * BK * 4);
    const unsigned b_lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)Bs + w * 32 * BK * 4);
    auto dma = [&](auto bufc, auto idxc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, IDX = decltype(idxc)::value;
        if constexpr (IDX < 2) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(a_lds0 + (BUF * TSA + 8 * IDX * BK) * 4), "v"(pa[IDX]) : "memory");
            pa[IDX] += BK;
        } else {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(b_lds0 + (BUF * TSB + 8 * (IDX - 2) * BK) * 4), "v"(pb[IDX - 2]) : "memory");
            pb[IDX - 2] += BK;
        }
        if constexpr (IDX == 5) {
            ck += BK;
            if (ck >= segK) {
                ck = 0;
                if (++cs < P.nseg) set_seg(cs);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int frow = lane & 31, khalf = lane >> 5;
    const int fsw = (frow >> 1) & 7;
    float4 fa[2], fb[2][TN];
    auto lfrag = [&](auto bufc, auto kbc, auto setc) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, kb = decltype(kbc)::value, ST = decltype(setc)::value;
        const int slot = ((2 * kb + khalf) ^ fsw) * 4;
        fa[ST] = *reinterpret_cast<const float4 *>(As + BUF * TSA + (wm * 32 + frow) * BK + slot);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fb[ST][j] = *reinterpret_cast<const float4 *>(Bs + BUF * TSB + ((wn * TN + j) * 32 + frow) * BK + slot);
    };
    auto mma_e = [&](auto setc, auto ec) __attribute__((always_inline)) {
        constexpr int ST = decltype(setc)::value, e = decltype(ec)::value;
        const float a[4] = {fa[ST].x, fa[ST].y, fa[ST].z, fa[ST].w};
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float b[4] = {fb[ST][j].x, fb[ST][j].y, fb[ST][j].z, fb[ST][j].w};
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[j], 0, 0, 0);
        }
    };
    auto mma = [&](auto setc) __attribute__((always_inline)) {
        mma_e(setc, I0{}); mma_e(setc, I1{}); mma_e(setc, I2{}); mma_e(setc, I3{});
    };
    int nchunks = 0;
    for (int s = 0; s < P.nseg; ++s) nchunks += P.seg[s].K / BK;
    auto chunk_body = [&](auto curc, auto nxtc, bool has1) __attribute__((always_inline)) {
        lfrag(curc, I1{}, I1{});
        mma_e(I0{}, I0{}); if (has1) { dma(nxtc, I0{}); dma(nxtc, I1{}); }
        mma_e(I0{}, I1{}); if (has1) { dma(nxtc, I2{}); dma(nxtc, I3{}); }
        mma_e(I0{}, I2{}); if (has1) { dma(nxtc, I4{}); dma(nxtc, I5{}); }
        mma_e(I0{}, I3{});
        lfrag(curc, I2{}, I0{});
        mma(I1{});
        lfrag(curc, I3{}, I1{});
        mma(I0{});
        if (has1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            lfrag(nxtc, I0{}, I0{});
        }
        mma(I1{});
    };
    set_seg(0);
    dma(I0{}, I0{}); dma(I0{}, I1{}); dma(I0{}, I2{}); dma(I0{}, I3{}); dma(I0{}, I4{}); dma(I0{}, I5{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    lfrag(I0{}, I0{}, I0{});
    for (int c = 0; c < nchunks; c += 2) {
        chunk_body(I0{}, I1{}, c + 1 < nchunks);
        if (c + 1 < nchunks) chunk_body(I1{}, I0{}, c + 2 < nchunks);
    }
    epi_linear_frag<TN>(P, acc, wm * 32, wn * 64, lane, row0, col0);
}


// ---------------------------------------------------------------- H3S: skinny split-f16 tiles for few-row launches
// (H3S_LAB_MODE = 1 / 2 / 3 builds time-only variants of the kernel below - DMA only / + fragment reads / MFMAs without
// reads - for tools/h3s_lab_modes.sh; 0, the default, compiles none of it.)
#ifndef H3S_LAB_MODE
#define H3S_LAB_MODE 0
#endif
// Launches with few rows (B <= a few hundred captions, beam rows, the 80-row seq2seq unroll) are not contraction-
// bound: each of their GEMMs streams a weight matrix far larger than its activations, and the step is a chain of
// such launches.  The fp32 route gave them a 32 x 128 tile per workgroup, split K over up to 16 workgroups, wrote
// [S, M, N] slabs and ran a second kernel to sum them and apply the epilogue (13 + 6 us per GEMM at M = 128).
// Here ONE launch does it.  Linear / LSTM: a workgroup owns a 32 x 32 output tile (LSTM: 8 hidden units x 4
// gates), so a [128 x 2048] launch spreads its weight stream over 256 workgroups, and its four waves take
// contiguous quarters of K; the four partial tiles meet in LDS, are summed in fixed order (deterministic) and
// the epilogue runs from the reduced tile.  Vocabulary projection: a 32 x 128 tile (its statistics are per 128
// columns), wave w owns columns 32w .. 32w+31 over the whole K.  Either way a wave is self-contained: it stages the
// 32-deep k-blocks of ITS operands - one [32 rows x 128 B] image each for A and W - into a private 4-slot LDS ring
// by LDS-DMA in pieces of 8 rows x 128 B (whole lines: a row's 32 hi + 32 lo halfs of a block are one line of the
// interleaved plane layout), three blocks in flight, ordered by its own vmcnt only: no barrier in the loop.
// LDS image: row r holds global 16-byte chunk c at position c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 fragments).
// Weights come as f16 planes from the stream's weights scope; an activation segment comes as planes when its
// producer wrote them, else as fp32 rows (same 128 bytes per row and block) and is split after the fragment read.
__device__ __forceinline__ void h3s_dma(unsigned lds_addr, const char *src) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_addr), "v"(src) : "memory");
}

// The 128-column statistics of a BM-row vocabulary tile from its four 32-column waves' (epi_vocab_frag<1, 4, BM>).
template <int BM>
__device__ __forceinline__ void h3s_vocab_combine(const DevProb &P, const float *smem, int tid, int row0, int tn) {
    const float *smx = smem;
    const float *ssm = smem + 4 * BM;
    const int *six = reinterpret_cast<const int *>(smem + 2 * 4 * BM);
    if (tid < BM && row0 + tid < P.M) {
        const int row = tid;
        float mx = smx[row];
        int ix = six[row];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ov = smx[w * BM + row];
            const int oi = six[w * BM + row];
            if (ov > mx || (ov == mx && oi < ix)) { mx = ov; ix = oi; }
        }
        float sm = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float wm_ = smx[w * BM + row];          // -inf: that wave had no valid column
            if (wm_ > -INFINITY) sm += ssm[w * BM + row] * __expf(wm_ - mx);
        }
        const long long o = (long long)(row0 + row) * P.ntile_total + tn;
        P.pmax[o] = mx;
        P.psum[o] = sm;
        P.pidx[o] = ix;
    }
}

// T = 2 (K-split epilogues only) doubles the tile to 64 x 64: a loaded row then feeds two MFMA tiles instead of one, so
// a launch needs half the bytes per output - the shape for launches of several rounds of 32 x 32 tiles (LSTM cell at
// M = 512 ... 1024) that are still too few 128-row tiles to fill the chip.  Its slots are 16 KB (64-row images), its
// ring two slots, and a slot is re-issued as soon as its fragments sit in registers (before the MFMAs), so two
// blocks - 32 KB per wave - stay in flight as in the 4-slot ring of the small tile.
template <int EPI, int T, int NW = 4>
__global__ __launch_bounds__(64 * NW) void gemm_h3s_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    rows_kernarg_warm<ROWS_KERNARG_LINES(DevLaunch)>();      // (one batch of scalar loads for the launch descriptor: common.h)
    constexpr bool KSPLIT = EPI != EPI_VOCAB;
    static_assert(T == 1 || (T == 2 && KSPLIT), "wide skinny tile: K-split epilogues only");
    static_assert(NW == 4 || (NW == 8 && T == 1), "eight waves: the small tiles only");
    constexpr int BM = 32 * T, BN = KSPLIT ? 32 * T : 128;
    // ring depth: 4 slots (three blocks in flight, 128 KB: one workgroup per CU) for the K-split tiles, whose launches
    // have at most ~256 workgroups; 2 slots (64 KB: two workgroups per CU) for the vocabulary projection, whose 79
    // column tiles x row tiles do not fit one round of single-workgroup CUs (316 workgroups at M = 128: 32 -> 17 us)
    // NW = 8: the K range over eight waves with two-slot rings (same 128 KB) - twice the waves issuing DMA
    constexpr int R = KSPLIT && T == 1 && NW == 4 ? 4 : 2, IMG = 4096 * T, SLOT = 2 * IMG, NP = 4 * T, LDR = BN + 1;
    constexpr bool EARLY = T == 2 || NW == 8;           // slot released after the fragment read, not after the MFMAs
    constexpr int LOGW = NW == 8 ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // 4 waves x R slots x (A image + W image)
    char *lds = reinterpret_cast<char *>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    // wave index as a scalar: everything derived from it (k-range, segment cursor) then lives in SGPRs and the
    // descriptor fields it selects are fetched by scalar loads, not by vector loads that would sit in vmcnt
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pi, tm, tn, ks, ksplit;
    map_tile(L, pi, tm, tn, ks, ksplit);
    const DevProb &P = L.p[pi];
    const int M = P.M, N = P.N, Kp = P.Kp;
    const int row0 = tm * BM, col0 = tn * BN;
    const int fr = lane & 31, fh = lane >> 5;
    // this wave's 32-column block inside the tile (vocabulary tile on eight waves: waves w and w + 4 share block w
    // and take one half of K each)
    const int cw = KSPLIT ? 0 : (wave & 3);

    // DMA lanes: piece p covers image rows 8p + (lane >> 3); lane position lane & 7 fetches chunk pos ^ swizzle(row)
    const int prow = lane >> 3;
    int a_row[NP];
    const char *w_src[NP];
    int chunk_off[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int r = 8 * p + prow;
        chunk_off[p] = ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        const int ar = row0 + r;
        a_row[p] = ar < M ? ar : M - 1;                 // rows past the edge fetch a valid row, never stored
        long long wr;
        if (EPI == EPI_LSTM) {                          // image row r = gate r / (8T), unit tn*8T + r % (8T)
            wr = (long long)(p / T) * P.H + tn * (8 * T) + (p % T) * 8 + prow;
        } else {
            const int c = col0 + cw * 32 + r;
            wr = c < N ? c : N - 1;
        }
        w_src[p] = reinterpret_cast<const char *>(P.Wh) + wr * 4 * Kp + chunk_off[p];
    }
    // this wave's k-blocks, and a cursor over the A segments positioned on its first block
    // ... of this workgroup's share of K (P.ksplit > 1: long contractions on few tiles are also split over workgroups;
    // the partial tiles then go to slabs and splitk_linear_kernel sums them and applies the epilogue)
    const int nblk = Kp >> 5;
    const int g_lo = (int)((long long)nblk * ks / ksplit), g_hi = (int)((long long)nblk * (ks + 1) / ksplit);
    const int nb = g_hi - g_lo;
    const int khalf = wave >> 2;                         // (vocabulary tile, NW = 8)
    const int b_lo = KSPLIT ? g_lo + ((nb * wave) >> LOGW) : (NW == 8 ? (nblk * khalf) >> 1 : 0);
    const int b_hi = KSPLIT ? g_lo + ((nb * (wave + 1)) >> LOGW) : (NW == 8 ? (nblk * (khalf + 1)) >> 1 : nblk);
    const int n = b_hi - b_lo;
    int cs = 0, cb = b_lo;
    while (cs < P.nap - 1 && cb >= (P.ap[cs].K >> 5)) { cb -= P.ap[cs].K >> 5; ++cs; }
    int seg_blocks = 0;
    bool seg_f32 = false;
    const char *a_src[NP];
    auto set_aseg = [&](int si) __attribute__((always_inline)) {
        const DevASeg a = P.ap[si];
        seg_blocks = a.K >> 5;
        seg_f32 = a.hi == nullptr;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (seg_f32)
                a_src[p] = reinterpret_cast<const char *>(P.seg[si].A + (long long)a_row[p] * P.seg[si].lda) + chunk_off[p];
            else
                a_src[p] = reinterpret_cast<const char *>(a.hi + (long long)a_row[p] * a.ld) + chunk_off[p];
        }
    };
    set_aseg(cs);
    const unsigned ring = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + wave * (R * SLOT));
    unsigned f32_bits = 0;                               // bit (block % R): that slot's A image holds fp32 rows
    int lb = b_lo;
    auto issue = [&](int i) __attribute__((always_inline)) {          // block i of this wave -> slot i % R
        const unsigned slot = ring + (i & (R - 1)) * SLOT;
        f32_bits = (f32_bits & ~(1u << (i & (R - 1)))) | ((seg_f32 ? 1u : 0u) << (i & (R - 1)));
#pragma unroll
        for (int p = 0; p < NP; ++p) h3s_dma(slot + p * 1024, a_src[p] + (long long)cb * 128);
#pragma unroll
        for (int p = 0; p < NP; ++p) h3s_dma(slot + IMG + p * 1024, w_src[p] + (long long)lb * 128);
        ++lb;
        if (++cb >= seg_blocks && cs + 1 < P.nap) { cb = 0; set_aseg(++cs); }
    };

    f32x16 acc0[T][T], acc1[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    const int fsw = (fr >> 1) & 7;
    constexpr int AHEAD = EARLY ? R : R - 1;             // blocks issued before the loop / kept ahead of the consumer
    for (int i = 0; i < AHEAD && i < n; ++i) issue(i);
    for (int i = 0; i < n; ++i) {
        if (!EARLY && i + R - 1 < n) issue(i + R - 1);   // its slot was consumed in iteration i - 1
        const int younger = n - 1 - i < R - 1 ? n - 1 - i : R - 1;    // blocks issued after block i, still in flight
        if (T == 1) {                                    // 8 DMA instructions per block
            if (R > 3 && younger == 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (R > 2 && younger == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {                                         // 16
            if (younger == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#if H3S_LAB_MODE == 1                                      // lab: DMA only (tools/h3s_lab_modes.sh; wrong results)
        if (EARLY) { if (i + R < n) issue(i + R); }
        continue;
#endif
        const char *img = lds + wave * (R * SLOT) + (i & (R - 1)) * SLOT + fr * 128;
        const bool af32 = (f32_bits >> (i & (R - 1))) & 1u;
        h8 a1[T][2], a2[T][2], b1[T][2], b2[T][2];
#if H3S_LAB_MODE == 3                                      // lab: MFMAs on constant fragments, no LDS reads
        for (int kk = 0; kk < 2; ++kk) for (int t = 0; t < T; ++t) for (int e = 0; e < 8; ++e) {
            a1[t][kk][e] = (_Float16)1.f; a2[t][kk][e] = (_Float16)1.f; b1[t][kk][e] = (_Float16)2.f; b2[t][kk][e] = (_Float16)3.f; }
        if (false)
#endif
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const char *ia = img + t * 4096;
                if (af32) {
                    const float4 v0 = *reinterpret_cast<const float4 *>(ia + (((4 * kk + 2 * fh) ^ fsw) * 16));
                    const float4 v1 = *reinterpret_cast<const float4 *>(ia + (((4 * kk + 2 * fh + 1) ^ fsw) * 16));
                    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const _Float16 h = (_Float16)x[e];
                        a1[t][kk][e] = h;
                        a2[t][kk][e] = (_Float16)((x[e] - (float)h) * 2048.f);
                    }
                } else {
                    a1[t][kk] = *reinterpret_cast<const h8 *>(ia + (((2 * kk + fh) ^ fsw) * 16));
                    a2[t][kk] = *reinterpret_cast<const h8 *>(ia + (((4 + 2 * kk + fh) ^ fsw) * 16));
                }
                const char *iw = img + IMG + t * 4096;
                b1[t][kk] = *reinterpret_cast<const h8 *>(iw + (((2 * kk + fh) ^ fsw) * 16));
                b2[t][kk] = *reinterpret_cast<const h8 *>(iw + (((4 + 2 * kk + fh) ^ fsw) * 16));
            }
        }
        if (EARLY) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the fragments are in registers: the slot is free
            if (i + R < n) issue(i + R);
        }
#if H3S_LAB_MODE == 2                                      // lab: DMA + fragment reads, no MFMAs
        for (int kk = 0; kk < 2; ++kk) for (int t = 0; t < T; ++t)
            asm volatile("" :: "v"(a1[t][kk]), "v"(a2[t][kk]), "v"(b1[t][kk]), "v"(b2[t][kk]));
        continue;
#endif
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ti = 0; ti < T; ++ti)
#pragma unroll
                for (int tj = 0; tj < T; ++tj) {
                    acc0[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ti][kk], b1[tj][kk], acc0[ti][tj], 0, 0, 0);
                    acc1[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ti][kk], b2[tj][kk], acc1[ti][tj], 0, 0, 0);
                    acc1[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[ti][kk], b1[tj][kk], acc1[ti][tj], 0, 0, 0);
                }
    }
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[i][j][r] = fmaf(acc1[i][j][r], 1.f / 2048.f, acc0[i][j][r]);
    __syncthreads();                                     // every wave is done with its ring: LDS is free

    if constexpr (EPI == EPI_VOCAB) {
        if (NW == 8) {                                   // upper-half-of-K partials -> the wave that owns the block
            float *xch = smem + (wave & 3) * (16 * 64);
            if (wave >= 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[r * 64 + lane] = acc0[0][0][r];
            }
            __syncthreads();
            if (wave < 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc0[0][0][r] += xch[r * 64 + lane];
            }
            __syncthreads();                             // the exchange area is the statistics' combine area next
        }
        f32x16 accv[1] = {acc0[0][0]};
        if (NW == 4 || wave < 4) epi_vocab_frag<1, 4, BM>(P, accv, 0, wave * 32, wave, lane, row0, col0, tn, smem);
        __syncthreads();
        h3s_vocab_combine<BM>(P, smem, tid, row0, tn);
        return;
    } else {
        // partial tile of this wave -> LDS (C/D layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
        float *mine = smem + wave * (BM * LDR);
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mine[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh) * LDR + j * 32 + fr] = acc0[i][j][r];
        __syncthreads();
        if (NW > 4 && tid >= 256) return;                // the epilogue maps 256 threads onto the tile
        auto red = [&](int row, int col) __attribute__((always_inline)) {
            const float *q = smem + row * LDR + col;
            float v = ((q[0] + q[BM * LDR]) + q[2 * BM * LDR]) + q[3 * BM * LDR];   // fixed order: bit-repeatable
            if (NW == 8) v += ((q[4 * BM * LDR] + q[5 * BM * LDR]) + q[6 * BM * LDR]) + q[7 * BM * LDR];
            return v;
        };
        if constexpr (EPI == EPI_LINEAR) {
#pragma unroll
            for (int q = 0; q < T * T; ++q) {
                const int grp = tid + 256 * q;                       // float4 group of the tile
                const int row = grp / (BN / 4), c4 = (grp % (BN / 4)) * 4;
                const int gm = row0 + row, gn = col0 + c4;
                if (gm >= M || gn >= N) continue;
                float o[4], pre[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = red(row, c4 + e);
                if (ksplit > 1) {                                    // raw partial sums (N % 4 == 0 on this route)
                    *reinterpret_cast<float4 *>(P.slab + (long long)ks * P.slab_stride + (long long)gm * N + gn) =
                        make_float4(o[0], o[1], o[2], o[3]);
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int nn = gn + e;
                    pre[e] = 0.f;
                    if (nn < N) {
                        if (P.bias0) o[e] += P.bias0[nn];
                        if (P.bias1) o[e] += P.bias1[nn];
                        if (P.bias2) o[e] += P.bias2[nn];
                        if (P.accumulate) o[e] += P.C[(long long)gm * P.ldc + nn];
                        if (P.relu) o[e] = isc_relu(o[e]);
                        pre[e] = o[e];
                        if (P.mask) o[e] = o[e] * (float)P.mask[(long long)gm * N + nn] * P.mask_scale;
                    }
                }
                float *dst = P.C + (long long)gm * P.ldc + gn;
                if ((P.ldc & 3) == 0 && gn + 3 < N) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                    if (P.C_pre)
                        *reinterpret_cast<float4 *>(P.C_pre + (long long)gm * P.ldc + gn) =
                            make_float4(pre[0], pre[1], pre[2], pre[3]);
                } else {
                    for (int e = 0; e < 4 && gn + e < N; ++e) {
                        dst[e] = o[e];
                        if (P.C_pre) P.C_pre[(long long)gm * P.ldc + gn + e] = pre[e];
                    }
                }
            }
        } else {
            constexpr int UN = 8 * T, NR = T * T;                    // units of the tile; rows per thread
            const int rbase = tid / UN, u = tid % UN;
            int gm[NR];
            bool ok[NR];
            float g[NR][4];
#pragma unroll
            for (int e = 0; e < NR; ++e) {
                const int row = rbase + e * (256 / UN);
                gm[e] = row0 + row;
                ok[e] = gm[e] < M;
#pragma unroll
                for (int k = 0; k < 4; ++k) g[e][k] = red(row, k * UN + u);
            }
            lstm_cells<NR>(P, gm, tn * UN + u, ok, g);
        }
    }
}

// ---------------------------------------------------------------- H3V: the vocabulary tile of few-hundred-row launches
// Same 32 x 128 tile, statistics and epilogue as gemm_h3s_kernel<EPI_VOCAB> (wave w owns columns 32 w .. 32 w + 31 over
// the whole K), different operand movement.  There every wave stages its own copy of the A block next to its W block in
// a private two-slot ring - A is fetched four times and ONE 8 KB block per wave is in flight: 64 KB per CU against a
// ~1.7 us round trip is ~9.5 TB/s over the chip, and 316 workgroups x 512 KB take the 20 us they are measured at
// (B = 128).  Here a k-block stage is shared by the workgroup: the A block once (each wave brings a quarter) + the four
// waves' W blocks = 20 KB, three stages (60 KB: two workgroups per CU), two blocks in flight behind the one being
// multiplied, one barrier per block: 320 KB per workgroup.  B = 128: 20.2 -> 18.2 us (kernel trace), roll-out 1.385 -> 1.32 ms.  (Tried first: W fragments straight from global memory into the
// MFMA operand registers, no LDS for W - a lane then touches 16 bytes of a 128-byte row line per instruction, the L1
// does not merge the four touches of a line, and the launch pulled ~4x its bytes from L2: 26 us.)
#ifndef H3V_STAGES
#define H3V_STAGES 3        // (four stages, 80 KB, measured the same: B = 128 roll-out 1.324 vs 1.323 ms)
#endif
#define H3V_STAGE_BYTES (5 * 4096)
template <bool AF32>
__global__ __launch_bounds__(256) void gemm_h3v_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    rows_kernarg_warm<ROWS_KERNARG_LINES(DevLaunch)>();
    constexpr int BM = 32, BN = 128;
    extern __shared__ __attribute__((aligned(16))) float smem[];     // stages of [A image 4 KB | 4 W images 4 KB each]
    char *lds = reinterpret_cast<char *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pi, tm, tn, ks, ksplit;
    map_tile(L, pi, tm, tn, ks, ksplit);
    const DevProb &P = L.p[pi];
    const int M = P.M, N = P.N, Kp = P.Kp, nblk = Kp >> 5;
    const int row0 = tm * BM, col0 = tn * BN;
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;

    // DMA lanes (pieces of 8 rows x 128 B; lane position lane & 7 fetches chunk pos ^ swizzle(row): the image layout of
    // gemm_h3s_kernel).  A: wave w brings rows 8 w .. 8 w + 7 of the block; W: the wave's own 32 columns, four pieces.
    const DevASeg a = P.ap[0];
    const bool af32 = AF32 && a.hi == nullptr;
    const char *asrc;
    const char *wsrc[4];
    {
        const int prow = 8 * wave + (lane >> 3);
        const int ar = row0 + prow < M ? row0 + prow : M - 1;
        const int choff = ((lane & 7) ^ ((prow >> 1) & 7)) * 16;
        asrc = af32 ? reinterpret_cast<const char *>(P.seg[0].A + (long long)ar * P.seg[0].lda) + choff
                    : reinterpret_cast<const char *>(a.hi + (long long)ar * a.ld) + choff;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = 8 * p + (lane >> 3);
            const int c = col0 + wave * 32 + r;
            wsrc[p] = reinterpret_cast<const char *>(P.Wh) + (long long)(c < N ? c : N - 1) * 4 * Kp +
                      ((lane & 7) ^ ((r >> 1) & 7)) * 16;
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds);
    auto issue = [&](int b, int stage) __attribute__((always_inline)) {       // 5 DMA instructions per wave
        const unsigned base = lds0 + stage * H3V_STAGE_BYTES;
        h3s_dma(base + wave * 1024, asrc + (long long)b * 128);
#pragma unroll
        for (int p = 0; p < 4; ++p) h3s_dma(base + 4096 + wave * 4096 + p * 1024, wsrc[p] + (long long)b * 128);
    };
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    constexpr int AHEAD = H3V_STAGES - 1;                  // blocks in flight behind the one being multiplied
    static_assert(AHEAD >= 1 && AHEAD <= 4, "stages");
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (i < nblk) issue(i, i);
    int st = 0;                                            // stage of block b
    for (int b = 0; b < nblk; ++b) {
        const int younger = nblk - 1 - b < AHEAD - 1 ? nblk - 1 - b : AHEAD - 1;    // blocks issued after block b
        if (younger >= 3) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // block b is in LDS for every wave; stage of block b - 1 is free
        if (b + AHEAD < nblk) issue(b + AHEAD, st == 0 ? H3V_STAGES - 1 : st - 1);
        const char *ia = lds + st * H3V_STAGE_BYTES + fr * 128;
        const char *iw = ia + 4096 + wave * 4096;
        h8 a1[2], a2[2], b1[2], b2[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (af32) {
                const float4 v0 = *reinterpret_cast<const float4 *>(ia + (((4 * kk + 2 * fh) ^ fsw) * 16));
                const float4 v1 = *reinterpret_cast<const float4 *>(ia + (((4 * kk + 2 * fh + 1) ^ fsw) * 16));
                const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const _Float16 h = (_Float16)x[e];
                    a1[kk][e] = h;
                    a2[kk][e] = (_Float16)((x[e] - (float)h) * 2048.f);
                }
            } else {
                a1[kk] = *reinterpret_cast<const h8 *>(ia + (((2 * kk + fh) ^ fsw) * 16));
                a2[kk] = *reinterpret_cast<const h8 *>(ia + (((4 + 2 * kk + fh) ^ fsw) * 16));
            }
            b1[kk] = *reinterpret_cast<const h8 *>(iw + (((2 * kk + fh) ^ fsw) * 16));
            b2[kk] = *reinterpret_cast<const h8 *>(iw + (((4 + 2 * kk + fh) ^ fsw) * 16));
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kk], b1[kk], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kk], b2[kk], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[kk], b1[kk], acc1, 0, 0, 0);
        }
        st = st == H3V_STAGES - 1 ? 0 : st + 1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = fmaf(acc1[r], 1.f / 2048.f, acc0[r]);
    __syncthreads();                                     // every wave is done with the stages: LDS is free
    f32x16 accv[1] = {acc0};
    epi_vocab_frag<1, 4, BM>(P, accv, 0, wave * 32, wave, lane, row0, col0, tn, smem);
    __syncthreads();
    h3s_vocab_combine<BM>(P, smem, tid, row0, tn);
}

// ---------------------------------------------------------------- split-K reduction + epilogue
// Sums the ksplit partial slabs in a fixed order (deterministic) and applies the epilogue the
// single-pass kernel would have applied.  blockIdx.y = problem.
__global__ __launch_bounds__(256) void splitk_linear_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    rows_kernarg_warm<ROWS_KERNARG_LINES(DevLaunch)>();      // (one batch of scalar loads for the launch descriptor: common.h)
    const DevProb &P = L.p[blockIdx.y];
    const long long n4 = (long long)P.M * (P.N >> 2);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int N = P.N, gm = (int)(i / (N >> 2)), gn = (int)(i % (N >> 2)) * 4;
    const float *sl = P.slab + (long long)gm * N + gn;
    float4 part[16];                            // every slab's float4 in flight before the first add
    const int S = P.ksplit;
#pragma unroll
    for (int s = 0; s < 16; ++s)
        part[s] = s < S ? *reinterpret_cast<const float4 *>(sl + (long long)s * P.slab_stride)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v = part[0];
#pragma unroll
    for (int s = 1; s < 16; ++s) { v.x += part[s].x; v.y += part[s].y; v.z += part[s].z; v.w += part[s].w; }
    float o[4] = {v.x, v.y, v.z, v.w}, pre[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = gn + e;
        if (P.bias0) o[e] += P.bias0[n];
        if (P.bias1) o[e] += P.bias1[n];
        if (P.bias2) o[e] += P.bias2[n];
        if (P.accumulate) o[e] += P.C[(long long)gm * P.ldc + n];
        if (P.relu) o[e] = isc_relu(o[e]);
        pre[e] = o[e];
        if (P.mask) o[e] = o[e] * (float)P.mask[(long long)gm * N + n] * P.mask_scale;
    }
    float *dst = P.C + (long long)gm * P.ldc + gn;
    if ((P.ldc & 3) == 0) {
        *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        if (P.C_pre)
            *reinterpret_cast<float4 *>(P.C_pre + (long long)gm * P.ldc + gn) = make_float4(pre[0], pre[1], pre[2], pre[3]);
    } else {
        for (int e = 0; e < 4; ++e) {
            dst[e] = o[e];
            if (P.C_pre) P.C_pre[(long long)gm * P.ldc + gn + e] = pre[e];
        }
    }
}

__global__ __launch_bounds__(256) void splitk_lstm_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    const DevProb &P = L.p[0];
    const int H = P.H;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)P.M * H) return;
    const int gm = (int)(i / H), unit = (int)(i % H);
    // all 4 x ksplit slab values are requested before the first add: summed one by one behind a runtime
    // trip count they were 16 dependent L2 round trips (15 us for a [4 x 2048] cell update)
    float part[4][16];
    const int S = P.ksplit;                   // <= 16 (plan_splitk)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float *sl = P.slab + (long long)gm * 4 * H + k * H + unit;
#pragma unroll
        for (int s = 0; s < 16; ++s) part[k][s] = s < S ? sl[(long long)s * P.slab_stride] : 0.f;
    }
    float g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a = part[k][0];
#pragma unroll
        for (int s = 1; s < 16; ++s) a += part[k][s];      // fixed order; the unused slots add 0
        if (P.bias0) a += P.bias0[k * H + unit] + P.bias1[k * H + unit];
        if (P.pre) a += P.pre[(long long)gm * 4 * H + k * H + unit];
        if (P.tab) a += P.tab[P.tab_ids[(long long)gm * P.tab_ids_stride] * 4 * H + k * H + unit];
        g[k] = a;
    }
    const float gi = isc_sigmoid(g[0]), gf = isc_sigmoid(g[1]), gg = isc_tanh(g[2]), go = isc_sigmoid(g[3]);
    const float c2 = gf * P.c_prev[i] + gi * gg;
    const float h2 = go * isc_tanh(c2);
    P.c_out[i] = c2;
    P.h_out[i] = h2;
    if (P.h_hi) {
        const _Float16 hh = (_Float16)h2;
        const long long po = plane_index(gm, unit, H);
        P.h_hi[po] = hh;
        P.h_lo[po] = (_Float16)((h2 - (float)hh) * 2048.f);
    }
    if (P.hmask) P.hdrop[i] = h2 * (float)P.hmask[i] * P.mask_scale;
    if (P.gates_out) {
        float *q = P.gates_out + (long long)gm * 4 * H + unit;
        q[0] = gi; q[H] = gf; q[2 * H] = gg; q[3 * H] = go;
    }
}

// Vocabulary projection by split-K (few rows): sums the slabs in fixed order, adds the bias, optionally
// writes the logits, and emits the per-128-column (max, arg-max, sum exp) the single-pass kernel's epilogue
// would have produced.  One wavefront per (row, column tile): lane l holds columns l and l + 64 of the tile.
__global__ __launch_bounds__(256) void splitk_vocab_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    const DevProb &P = L.p[0];
    const int lane = threadIdx.x & 63;
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_tile = P.ntile_total;
    if (w >= (long long)P.M * n_tile) return;
    const int gm = (int)(w / n_tile), tn = (int)(w % n_tile);
    const int N = P.N, S = P.ksplit;
    float x[2];
    int col[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        col[h] = tn * 128 + lane + 64 * h;
        const bool ok = col[h] < N;
        const float *sl = P.slab + (long long)gm * N + (ok ? col[h] : 0);
        float part[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) part[s] = (ok && s < S) ? sl[(long long)s * P.slab_stride] : 0.f;
        float a = part[0];
#pragma unroll
        for (int s = 1; s < 16; ++s) a += part[s];
        x[h] = ok ? a + P.bias0[col[h]] : -INFINITY;
        if (ok && P.C) P.C[(long long)gm * P.ld_logits + col[h]] = x[h];
    }
    float mx = x[0];
    int ix = col[0];
    if (x[1] > mx) { mx = x[1]; ix = col[1]; }             // equal: the lower column (h = 0) stays
    wave_argmax(mx, ix);
    float sm = (x[0] > -INFINITY ? __expf(x[0] - mx) : 0.f) + (x[1] > -INFINITY ? __expf(x[1] - mx) : 0.f);
    sm = wave_sum(sm);
    if (lane == 0) {
        const long long o = (long long)gm * n_tile + tn;
        P.pmax[o] = mx;
        P.psum[o] = sm;
        P.pidx[o] = ix;
    }
}

// ---------------------------------------------------------------- few rows: fused GEMV kernels (exact fp32)
// Beam rows of one image (5), roll-outs of a handful of captions: with M <= 8 rows a "GEMM" is M matrix-vector
// products that stream the weights once - no tile shape reuses anything, the 32-row MFMA tiles of the skinny kernel
// are 84-97 % padding, and its launches sit at the latency of their 8-block DMA chains (LSTM cell at 5 rows: 9.4 us
// for 16.8 MB of weight planes = 1.8 TB/s).  Here the weights are read as they are (fp32, no planes, no split - the
// arithmetic is exact fp32 FMA) by wide per-lane loads with the whole launch's traffic in flight at once:
//   * a workgroup (4 waves) stages the M activation rows [M, K] of the launch in LDS (fp32, all segments, <= 128 KB);
//   * a wave owns 4 weight rows at a time (linear / vocabulary: 4 consecutive output columns; LSTM: the i, f, g, o
//     rows of one hidden unit); lane l takes the float4 at k = 256 c + 4 l of each row for up to 8 chunks c - 32
//     loads of 16 B per lane in flight (the vocabulary tile: 8 rows x 2 chunks) - and accumulates 4 x M dot products;
//   * the 4 M (<= 32) per-lane sums are reduced across the 64 lanes by a reduce-scatter butterfly (each step halves
//     the values a lane carries: 31 shuffles instead of 32 x 6), fixed order, deterministic;
//   * epilogues from LDS: linear (biases, accumulate, ReLU, keep-mask, pre-mask copy), LSTM cell (lstm_cells<1>: the
//     same code as the tile kernels - hoisted term, token-table row, planes of h for consumers, saved gates),
//     vocabulary (logits + the per-128-column max / arg-max / sum-exp statistics of the tile kernels' epilogue).
// north_star's "fused LSTM gate GEMV + sigmoid/tanh" is this kernel's <EPI_LSTM> form.
#define GEMV_MAX_ROWS 8
#define GEMV_MAX_K 4096
// v[0 .. NV) per lane -> after the call v[0] of lane l holds the 64-lane sum of value index l / (64 / NV).
template <int HALF, int MASK, int NV>
__device__ __forceinline__ void wave_reduce_scatter_step(float (&v)[NV], int lane) {
    const bool up = (lane & MASK) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
        const float send = up ? v[i] : v[i + HALF];
        const float keep = up ? v[i + HALF] : v[i];
        v[i] = keep + __shfl_xor(send, MASK, 64);
    }
    if constexpr (HALF > 1) wave_reduce_scatter_step<HALF / 2, MASK / 2, NV>(v, lane);
}
template <int NV>
__device__ __forceinline__ void wave_reduce_scatter(float (&v)[NV], int lane) {
    static_assert(NV == 32 || NV == 64, "4 or 8 weight rows x 8 activation rows");
    wave_reduce_scatter_step<NV / 2, 32, NV>(v, lane);
    if (NV == 32) v[0] += __shfl_xor(v[0], 1, 64);
}

// One batch of CB 256-wide k-chunks: NR x CB unconditional 16-byte loads per lane (a lane past the segment's end reads
// the row's last float4 again and meets a zero activation: no branch sits between a load and its use, so the loads
// stay in flight together - a predicated load gets its own s_waitcnt), then the FMAs.
template <int NR, int CB>
__device__ __forceinline__ void gemv_batch(const float *Wb, const long long (&wrow)[NR], long long ldw, int Ks, int c0,
                                           const float *Ak, int Ktot, int lane, float (&acc)[NR * GEMV_MAX_ROWS]) {
    constexpr int MR = GEMV_MAX_ROWS;
    float4 w[NR][CB];
    int kc[CB];
    bool in[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
        const int kk = c0 + c * 256 + lane * 4;
        in[c] = kk < Ks;
        kc[c] = in[c] ? kk : Ks - 4;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < CB; ++c) w[r][c] = *reinterpret_cast<const float4 *>(Wb + wrow[r] * ldw + kc[c]);
#pragma unroll
    for (int c = 0; c < CB; ++c) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float4 a = *reinterpret_cast<const float4 *>(Ak + (long long)m * Ktot + kc[c]);
            a.x = in[c] ? a.x : 0.f; a.y = in[c] ? a.y : 0.f; a.z = in[c] ? a.z : 0.f; a.w = in[c] ? a.w : 0.f;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float t = acc[r * MR + m];
                t = fmaf(w[r][c].x, a.x, t);
                t = fmaf(w[r][c].y, a.y, t);
                t = fmaf(w[r][c].z, a.z, t);
                t = fmaf(w[r][c].w, a.w, t);
                acc[r * MR + m] = t;
            }
        }
    }
}

// NQ x 4 weight rows against the MR staged activation rows over this wave's share of K: chunks [g_lo, g_hi) of the
// launch's chunk list (the segments' 256-wide chunks, concatenated); at most CBMAX chunks per batch.
template <int NQ, int CBMAX>
__device__ __forceinline__ void gemv_accumulate(const DevProb &P, const long long (&wrow)[NQ * 4], const float *As,
                                                int Ktot, int lane, int g_lo, int g_hi,
                                                float (&acc)[NQ * 4 * GEMV_MAX_ROWS]) {
    constexpr int NR = NQ * 4, MR = GEMV_MAX_ROWS;
#pragma unroll
    for (int i = 0; i < NR * MR; ++i) acc[i] = 0.f;
    int koff = 0, g0 = 0;
    for (int s = 0; s < P.nseg; ++s) {
        const int Ks = P.seg[s].K, nch = (Ks + 255) >> 8;
        const float *Wb = P.seg[s].W;
        const long long ldw = P.seg[s].ldw;
        const float *Ak = As + koff;
        const int lo = g_lo > g0 ? g_lo - g0 : 0, hi = g_hi - g0 < nch ? g_hi - g0 : nch;   // this segment's chunks (uniform)
        int c0 = lo * 256, rem = hi - lo;
        if constexpr (CBMAX >= 8) {
            while (rem >= 8) { gemv_batch<NR, 8>(Wb, wrow, ldw, Ks, c0, Ak, Ktot, lane, acc); c0 += 2048; rem -= 8; }
        }
        if constexpr (CBMAX >= 4) {
            while (rem >= 4) { gemv_batch<NR, 4>(Wb, wrow, ldw, Ks, c0, Ak, Ktot, lane, acc); c0 += 1024; rem -= 4; }
        }
        while (rem >= 2) { gemv_batch<NR, 2>(Wb, wrow, ldw, Ks, c0, Ak, Ktot, lane, acc); c0 += 512; rem -= 2; }
        if (rem >= 1) gemv_batch<NR, 1>(Wb, wrow, ldw, Ks, c0, Ak, Ktot, lane, acc);
        koff += Ks;
        g0 += nch;
    }
}

// NW waves per workgroup.  Linear / LSTM (NW = 4): a tile is 4 / KS quads (4 output columns, or the 4 gate rows of a
// hidden unit), each contracted by KS waves over 1 / KS of the chunk list (P.ksplit = KS in {1, 2, 4}: long contractions
// spread over more waves so that every wave's loads fit one batch and the launch fills the chip - the LSTM cell over
// K = 2048 at KS = 2 is 256 workgroups of 4 waves with 16 loads each); the KS partial sums meet in LDS in fixed order.
// Vocabulary (NW = 8): a tile is 128 columns (its statistics are per 128), wave w takes columns 16 w .. 16 w + 15 in
// two rounds of 8 rows.
template <int EPI, int NW>
__global__ __launch_bounds__(64 * NW) void gemv_rows_kernel(const DevLaunch L) {
    ISC_GATE_RETURN(L);
    rows_kernarg_warm<ROWS_KERNARG_LINES(DevLaunch)>();      // (one batch of scalar loads for the launch descriptor: common.h)
    constexpr int MR = GEMV_MAX_ROWS, NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pi = 0;
    if (L.nprob > 1 && (int)blockIdx.x >= L.p[1].tile_start) pi = 1;
    if (L.nprob > 2 && (int)blockIdx.x >= L.p[2].tile_start) pi = 2;
    const DevProb &P = L.p[pi];
    const int tile = blockIdx.x - P.tile_start;
    const int M = P.M, N = P.N;
    int Ktot = 0, nchunk = 0;
    for (int s = 0; s < P.nseg; ++s) { Ktot += P.seg[s].K; nchunk += (P.seg[s].K + 255) >> 8; }
    float *As = smem;                                   // [MR][Ktot]
    float *red = smem + (long long)MR * Ktot;           // linear / LSTM: [4 waves][32]; vocabulary: [MR][128]
    {                                                   // stage the activations (rows >= M: zeros)
        int koff = 0;
        for (int s = 0; s < P.nseg; ++s) {
            const int Ks = P.seg[s].K, q4 = Ks >> 2;
            const float *Ab = P.seg[s].A;
            const long long lda = P.seg[s].lda;
            for (int k4 = tid; k4 < q4; k4 += NT) {     // all MR row loads of a column in flight (rows >= M: row M - 1)
                float4 v[MR];
#pragma unroll
                for (int m = 0; m < MR; ++m)
                    v[m] = *reinterpret_cast<const float4 *>(Ab + (long long)(m < M ? m : M - 1) * lda + k4 * 4);
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    if (m >= M) v[m] = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4 *>(As + (long long)m * Ktot + koff + k4 * 4) = v[m];
                }
            }
            koff += Ks;
        }
    }
    __syncthreads();

    if constexpr (EPI == EPI_VOCAB) {
        const int col0 = tile * 128;
        for (int rnd = 0; rnd < 2; ++rnd) {
            long long wrow[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int c = col0 + wave * 16 + rnd * 8 + r;
                wrow[r] = c < N ? c : N - 1;
            }
            float acc[8 * MR];
            gemv_accumulate<2, 2>(P, wrow, As, Ktot, lane, 0, nchunk, acc);
            wave_reduce_scatter<64>(acc, lane);          // lane l: value (row r = l / 8 of the round, m = l % 8)
            red[(lane & 7) * 128 + wave * 16 + rnd * 8 + (lane >> 3)] = acc[0];
        }
        __syncthreads();
        const int n_tile = P.ntile_total;
        for (int m = wave; m < M; m += NW) {
            float x[2];
            int col[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                col[h] = col0 + lane + 64 * h;
                const bool ok = col[h] < N;
                x[h] = ok ? red[m * 128 + lane + 64 * h] + P.bias0[col[h]] : -INFINITY;
                if (ok && P.C) P.C[(long long)m * P.ld_logits + col[h]] = x[h];
            }
            float mx = x[0];
            int ix = col[0];
            if (x[1] > mx) { mx = x[1]; ix = col[1]; }             // equal: the lower column stays
            wave_argmax(mx, ix);
            float sm = (x[0] > -INFINITY ? __expf(x[0] - mx) : 0.f) + (x[1] > -INFINITY ? __expf(x[1] - mx) : 0.f);
            sm = wave_sum(sm);
            if (lane == 0) {
                const long long o = (long long)m * n_tile + tile;
                P.pmax[o] = mx;
                P.psum[o] = sm;
                P.pidx[o] = ix;
            }
        }
    } else {
        const int KS = P.ksplit;                         // 1, 2 or 4 (try_gemv)
        const int qpw = NW / KS;                         // quads per workgroup
        const int qw = wave / KS, ks = wave - qw * KS;   // this wave: quad qw of the tile, k-slice ks
        const int quad = tile * qpw + qw;
        long long wrow[4];
        bool live;
        if constexpr (EPI == EPI_LSTM) {
            live = quad < P.H;
#pragma unroll
            for (int r = 0; r < 4; ++r) wrow[r] = (long long)r * P.H + (live ? quad : 0);
        } else {
            live = quad * 4 < N;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int c = quad * 4 + r; wrow[r] = c < N ? c : N - 1; }
        }
        float acc[4 * MR];
        gemv_accumulate<1, 8>(P, wrow, As, Ktot, lane, nchunk * ks / KS, nchunk * (ks + 1) / KS, acc);
        wave_reduce_scatter<32>(acc, lane);              // lanes 2 i, 2 i + 1: value i = (row r = i / 8, m = i % 8)
        if ((lane & 1) == 0) red[wave * 32 + (lane >> 1)] = acc[0];
        __syncthreads();
        if (ks == 0 && live) {                           // the quad's first wave sums the KS partials in order
            const float *rq = red + wave * 32;
            if constexpr (EPI == EPI_LSTM) {
                if (lane < M) {
                    const int gm[1] = {lane};
                    const bool ok[1] = {true};
                    float g[1][4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float v = rq[k * MR + lane];
                        for (int j = 1; j < KS; ++j) v += rq[j * 32 + k * MR + lane];
                        g[0][k] = v;
                    }
                    lstm_cells<1>(P, gm, quad, ok, g);
                }
            } else {
                const int m = lane >> 2, e = lane & 3, n = quad * 4 + e;
                if (lane < 4 * MR && m < M && n < N) {
                    float o = rq[e * MR + m];
                    for (int j = 1; j < KS; ++j) o += rq[j * 32 + e * MR + m];
                    if (P.bias0) o += P.bias0[n];
                    if (P.bias1) o += P.bias1[n];
                    if (P.bias2) o += P.bias2[n];
                    if (P.accumulate) o += P.C[(long long)m * P.ldc + n];
                    if (P.relu) o = isc_relu(o);
                    if (P.C_pre) P.C_pre[(long long)m * P.ldc + n] = o;
                    if (P.mask) o = o * (float)P.mask[(long long)m * N + n] * P.mask_scale;
                    P.C[(long long)m * P.ldc + n] = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- host side
static int check_segs(const isc_seg *seg, int nseg) {
    if (nseg < 1 || nseg > ISC_MAX_SEG) return ISC_E_SHAPE;
    for (int s = 0; s < nseg; ++s) {
        if (!seg[s].A || !seg[s].W) return ISC_E_NULL;
        if (seg[s].K <= 0 || (seg[s].K % BK) != 0) return ISC_E_SHAPE;
        if ((seg[s].lda & 3) || (seg[s].ldw & 3)) return ISC_E_ALIGN;
        if (!isc_aligned16(seg[s].A) || !isc_aligned16(seg[s].W)) return ISC_E_ALIGN;
    }
    return ISC_OK;
}

static void copy_segs(DevProb &d, const isc_seg *seg, int nseg) {
    d.nseg = nseg;
    for (int s = 0; s < nseg; ++s) {
        d.seg[s].A = seg[s].A; d.seg[s].W = seg[s].W;
        d.seg[s].lda = seg[s].lda; d.seg[s].ldw = seg[s].ldw; d.seg[s].K = seg[s].K; d.seg[s].pad = 0;
        d.seg[s].A_hi = static_cast<const _Float16 *>(seg[s].A_hi);
        d.seg[s].A_lo = static_cast<const _Float16 *>(seg[s].A_lo);
    }
}

template <int WM, int WN, int TN, int EPI, bool AKM, bool BKM>
static int launch_cfg(const DevLaunch &L, hipStream_t st) {
    constexpr int BM = 32 * WM, BN = 32 * TN * WN;
    constexpr size_t k_bytes = (size_t)2 * (Tile<BM, AKM>::SIZE + Tile<BN, BKM>::SIZE) * sizeof(float);
    constexpr size_t c_bytes = (size_t)BM * (BN + 4) * sizeof(float);
    constexpr size_t lds = k_bytes > c_bytes ? k_bytes : c_bytes;
    static std::atomic<bool> attr_set{false};  // idempotent: a race only repeats the same call
    if (!attr_set.load() && lds > 65536) {
        hipError_t e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&gemm_kernel<WM, WN, TN, EPI, AKM, BKM>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set.store(true);
    }
    hipLaunchKernelGGL((gemm_kernel<WM, WN, TN, EPI, AKM, BKM>), dim3(L.total_tiles), dim3(256), lds, st, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

template <int EPI>
static int launch_xl(const DevLaunch &L, hipStream_t st) {
    constexpr size_t lds = (size_t)3 * (256 + 128) * BK * sizeof(float);   // 147456 >= Cs 256 x 132 floats
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_xl_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set.store(true);
    }
    hipLaunchKernelGGL((gemm_xl_kernel<EPI>), dim3(L.total_tiles), dim3(256), lds, st, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

template <int EPI>
static int launch_ld(const DevLaunch &L, hipStream_t st) {
    constexpr size_t lds = (size_t)4 * 128 * BK * sizeof(float);           // 65536: two workgroups per CU
    hipLaunchKernelGGL((gemm_ld_kernel<EPI>), dim3(L.total_tiles), dim3(256), lds, st, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// Tile shapes: 0 = L 128x128 (4 waves stacked in M), 1 = M 64x128 (2x2 waves), 2 = S 32x128 (4 waves in N),
// 3 = XL 256x128 (LDS-DMA ring, NT layout only), 4 = LD 128x128 by LDS-DMA (NT layout only)
static const int kTileBM[5] = {128, 64, 32, 256, 128};

static int launch_md(const DevLaunch &L, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * (64 + 128) * BK * sizeof(float);    // 49152: three workgroups per CU
    hipLaunchKernelGGL(gemm_md_kernel, dim3(L.total_tiles), dim3(256), lds, st, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

static std::atomic<int> g_md_enabled{1};   // isc_set_tile_override(102 / 103): MD path off / on (A/B measurements)

template <int EPI, bool AKM, bool BKM>
static int launch_any(const DevLaunch &L, int tile, hipStream_t st) {
    if constexpr (EPI == EPI_LINEAR && !AKM && !BKM) {
        if (tile == 1 && g_md_enabled.load()) {
            bool plain = true;
            for (int i = 0; i < L.nprob; ++i) plain = plain && L.p[i].ksplit <= 1;
            if (plain) return launch_md(L, st);
        }
    }
    if constexpr (!AKM && !BKM) {
        if (tile == 3) return launch_xl<EPI>(L, st);
        if (tile == 4) return launch_ld<EPI>(L, st);
    }
    if (tile == 0) return launch_cfg<4, 1, 4, EPI, AKM, BKM>(L, st);
    if (tile == 1) return launch_cfg<2, 2, 2, EPI, AKM, BKM>(L, st);
    return launch_cfg<1, 4, 1, EPI, AKM, BKM>(L, st);
}

// Tile shape choice by a small cost model.  The most loaded CU works through n = ceil(blocks / 256)
// workgroups; its time is n * BM / efficiency, where the efficiency depends on the tile (MFMAs per
// staged byte and per barrier) and on how many workgroups share the CU (the 128/64/32-row tiles rely
// on a co-resident workgroup to cover their chunk-boundary stalls, the XL tile is pipelined to run
// alone).  Calibrated on tools/gemm_big.py / gemm_bench.py.
// Process-wide tuning knobs (tile override, split-f16 mode): plain configuration words, read once per launch
// (atomics: flipping them from another thread is a data-race-free, if pointless, thing to do).  They select a
// kernel; they carry no data between calls.
static std::atomic<int> g_tile_override{-1};
extern "C" int isc_set_tile_override(int tile) {
    if (tile == 102 || tile == 103) {          // measurement switch for the MD path, leaves the tile choice alone
        g_md_enabled.store(tile == 103);
        return g_tile_override.load();
    }
    return g_tile_override.exchange((tile >= 0 && tile <= 4) ? tile : -1);
}

// nt: the launch is in the NT layout (the LDS-DMA tiles XL / LD exist for it; MD replaces M for the linear epilogue
// inside launch_any); use_ld: let the cost model consider the LD tile
static int pick_tile(const DevLaunch &L, bool allow_xl, bool nt = true, bool use_ld = true) {
    const int ovr = g_tile_override.load();
    if (ovr == 4) return nt ? 4 : 0;
    if (ovr >= 0) return (ovr == 3 && !allow_xl) ? 0 : ovr;
    static const double eff[5][3] = {{0.80, 1.00, 1.00},    // L : 1, 2, >=3 workgroups on the busiest CU
                                     {0.55, 0.80, 0.90},    // M
                                     {0.35, 0.50, 0.62},    // S
                                     {1.10, 1.10, 1.10},    // XL
                                     {0.85, 1.06, 1.12}};   // LD
    int best = 0;
    double best_cost = 1e30;
    static const int order[5] = {3, 4, 0, 1, 2};            // larger / DMA tiles first: they win near-ties
    for (int k = 0; k < 5; ++k) {
        const int t = order[k];
        if (t == 3 && !allow_xl) continue;
        if (t == 4 && !(nt && use_ld)) continue;
        long long blocks = 0;
        for (int i = 0; i < L.nprob; ++i)
            blocks += (long long)((L.p[i].M + kTileBM[t] - 1) / kTileBM[t]) * ((L.p[i].N + 127) / 128);
        const long long n = (blocks + 255) / 256;
        const double cost = (double)n * kTileBM[t] / eff[t][n >= 3 ? 2 : (int)n - 1];
        if (cost < best_cost * 0.97) { best_cost = cost; best = t; }
    }
    return best;
}

static void finish_tiling(DevLaunch &L, int tile) {
    const int BM = kTileBM[tile], BN = 128;
    int start = 0;
    for (int i = 0; i < L.nprob; ++i) {
        DevProb &p = L.p[i];
        p.tiles_m = (p.M + BM - 1) / BM;
        p.tiles_n = (p.N + BN - 1) / BN;
        p.tile_start = start;
        long long wbytes = 0, abytes = 0;
        for (int s = 0; s < p.nseg; ++s) { wbytes += (long long)p.N * p.seg[s].K; abytes += (long long)p.M * p.seg[s].K; }
        p.m_fastest = wbytes > abytes;  // partition the larger operand across XCDs
        p.grp_n = (p.tiles_n + 7) / 8;
        // ... or the smaller one when the fabric-traffic estimate of the grouped order is under half of row-major's: the C workgroups an XCD runs at a
        // time form an a x b block of tiles and fetch a row panels + b column panels per round; a weight slice that
        // fits the XCD's L2 is fetched once.  (The classifier at B = 16384 has the larger A, and row-major order put
        // 64 different weight panels per round through each L2: 2.7 GB per launch, r03_e PMC; grouped: 8 A + W = 0.3 GB.)
        {
            const double Ap = 4.0 * BM * (double)(abytes / (p.M > 0 ? p.M : 1)), Wp = 4.0 * BN * (double)(wbytes / (p.N > 0 ? p.N : 1));
            const double A = 4.0 * abytes, W = 4.0 * wbytes, L2 = 3.0 * 1048576.0;
            const int C = BM >= 256 ? 32 : 64;
            const double rounds = (double)p.tiles_m * p.tiles_n / (8.0 * C);
            const int b = p.tiles_n < C ? p.tiles_n : C, a = C / p.tiles_n > 1 ? C / p.tiles_n : 1;
            const double row = W <= L2 ? A + 8.0 * W : 8.0 * rounds * (a * Ap + b * Wp);
            const int gb = p.grp_n < C ? p.grp_n : C, ga = C / p.grp_n > 1 ? C / p.grp_n : 1;
            const double grp = p.grp_n * Wp <= L2 ? 8.0 * A + W : 8.0 * rounds * (ga * Ap + gb * Wp);
            if (!p.m_fastest && grp < 0.5 * row) p.m_fastest = 1;   // (the opposite switch measured worse at B = 128)
        }
        start += p.tiles_m * p.tiles_n * (p.ksplit > 1 ? p.ksplit : 1);
    }
    L.total_tiles = start;
}

// ---- split-f16 path (gemm_h3_kernel) ----
// isc_set_h3_mode: 0 = off, 1 = auto, 2 = large kernels whenever the shapes allow, 3 / 4 = skinny kernel (32 / 64 tiles)
static std::atomic<int> g_h3_mode{1};
static std::atomic<long long> g_h3_launches{0}, g_h3x_launches{0};
extern "C" long long isc_h3_launches(void) { return g_h3_launches.load(); }
extern "C" long long isc_h3x_launches(void) { return g_h3x_launches.load(); }
extern "C" int isc_set_h3_mode(int mode) {
    if (mode >= 0 && mode <= 4) return g_h3_mode.exchange(mode);
    return g_h3_mode.load();
}
static std::atomic<int> g_gemv_rows{GEMV_MAX_ROWS};
static std::atomic<long long> g_gemv_launches{0};
extern "C" int isc_set_gemv_rows(int rows) {           // 0 = off; returns the previous value
    if (rows >= 0 && rows <= GEMV_MAX_ROWS) return g_gemv_rows.exchange(rows);
    return g_gemv_rows.load();
}
extern "C" long long isc_gemv_launches(void) { return g_gemv_launches.load(); }

// Takes the launch when every problem has <= g_gemv_rows rows and fits the LDS image.  Returns 1 when it went out.
template <int EPI>
static int try_gemv(DevLaunch &L, hipStream_t st, int &rc) {
    constexpr int NW = EPI == EPI_VOCAB ? 8 : 4;
    const int rows = g_gemv_rows.load(), mode = g_h3_mode.load();
    if (rows <= 0 || mode > 1 || g_tile_override.load() >= 0) return 0;
    int start = 0, kmax = 0;
    for (int i = 0; i < L.nprob; ++i) {
        DevProb &p = L.p[i];
        if (p.M > rows || p.ksplit > 1) return 0;
        int K = 0, nchunk = 0;
        for (int s = 0; s < p.nseg; ++s) {
            if ((p.seg[s].K & 3) || (p.seg[s].lda & 3) || (p.seg[s].ldw & 3)) return 0;
            K += p.seg[s].K;
            nchunk += (p.seg[s].K + 255) >> 8;
        }
        if (K > GEMV_MAX_K) return 0;
        if (EPI == EPI_LSTM && (p.H <= 0 || p.N != 4 * p.H)) return 0;
        kmax = K > kmax ? K : kmax;
    }
    for (int i = 0; i < L.nprob; ++i) {
        DevProb &p = L.p[i];
        int nchunk = 0;
        for (int s = 0; s < p.nseg; ++s) nchunk += (p.seg[s].K + 255) >> 8;
        // K slices inside the workgroup: at most 4 chunks (16 loads) per wave, and enough workgroups for the chip
        const int quads = EPI == EPI_LSTM ? p.H : (p.N + 3) / 4;
        int KS = 1;
        if (EPI != EPI_VOCAB) {       // the smallest of 1, 2, 4 with <= 4 chunks per wave and >= 200 workgroups
            while (KS < 4 && KS * 2 <= nchunk && ((nchunk + KS - 1) / KS > 4 || (long long)quads * KS / NW < 200)) KS *= 2;
        }
        p.ksplit = KS;
        p.tile_start = start;
        start += EPI == EPI_VOCAB ? (p.N + 127) / 128 : (quads + NW / KS - 1) / (NW / KS);
    }
    L.total_tiles = start;
    const size_t lds = ((size_t)GEMV_MAX_ROWS * kmax + (EPI == EPI_VOCAB ? GEMV_MAX_ROWS * 128 : NW * 32)) * sizeof(float);
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemv_rows_kernel<EPI, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(((size_t)GEMV_MAX_ROWS * GEMV_MAX_K + GEMV_MAX_ROWS * 128) * sizeof(float)));
        if (e != hipSuccess) { rc = (int)e; return 1; }
        attr_set.store(true);
    }
    hipLaunchKernelGGL((gemv_rows_kernel<EPI, NW>), dim3(L.total_tiles), dim3(64 * NW), lds, st, L);
    rc = ISC_OK;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = (int)e;
    for (int i = 0; i < L.nprob; ++i) L.p[i].ksplit = 0;
    ++g_gemv_launches;
    return 1;
}

#define H3_MIN_TILES 160
#define H3_MIN_TILES_SCOPE 16
// (tests / A-B runs: 0 keeps long contractions on few large tiles in one slice of K)
static std::atomic<int> g_h3_ksplit{1};
extern "C" int isc_set_h3_ksplit(int on) { return g_h3_ksplit.exchange(on < 0 ? 0 : (on > 2 ? 2 : on)); }
static int launch_splitk_linear_reduce(const DevLaunch &L, hipStream_t st);

static bool h3_any_f32(const DevLaunch &L) {
    for (int i = 0; i < L.nprob; ++i)
        for (int s = 0; s < L.p[i].nap; ++s)
            if (!L.p[i].ap[s].hi) return true;
    return false;
}

template <int EPI>
static int launch_h3(const DevLaunch &L, hipStream_t st) {
    constexpr size_t lds = 65536;                                          // two workgroups per CU
    if (h3_any_f32(L)) hipLaunchKernelGGL((gemm_h3_kernel<EPI, true>), dim3(L.total_tiles), dim3(256), lds, st, L);
    else hipLaunchKernelGGL((gemm_h3_kernel<EPI, false>), dim3(L.total_tiles), dim3(256), lds, st, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

static int h3_kp(const DevProb &p) {
    int Kp = 0;
    for (int s = 0; s < p.nseg; ++s) Kp += p.seg[s].K;
    return Kp;
}

// Weight planes kept across launches while the caller guarantees the weights do not change
// (isc_h3_weights_begin / _end, include/insenticap_hip.h): a roll-out splits each weight matrix once, not once
// per step.  Host-side table, valid on the stream order of the launches that filled it.
struct H3WEntry {
    const float *W[ISC_MAX_SEG];
    int ldw[ISC_MAX_SEG], K[ISC_MAX_SEG], nseg, rows, transposed;
    const _Float16 *hi, *lo;
};
// One scope per stream (isc_h3_weights_begin(buf, bytes, stream)): the slot table is guarded by a mutex, a slot's
// entries are only touched by launches on its own stream (which the caller issues in one order, like the launches
// themselves), so two host threads driving two captioners on two streams never see each other's planes.
#define H3W_MAX_ENTRIES 72
struct H3WScope {
    hipStream_t stream = nullptr;
    char *buf = nullptr;
    size_t bytes = 0, used = 0;
    H3WEntry e[H3W_MAX_ENTRIES];
    int n = 0;
    bool active = false;
};
// 64 slots: more than the streams a process can hold per device (torch hands out 32 pool streams per device and priority,
// plus the default stream); begin never takes over a SUSPENDED slot of another stream.  With 16 slots a test session
// that kept many captioners / training graphs alive ran out of them, and after such a take-over a captured training
// graph (seq2seq branch) computed its loss from planes one update old (graph vs eager 6.0207 / 6.0188; order-dependent).
#define H3W_MAX_SCOPES 64
static H3WScope g_h3w[H3W_MAX_SCOPES];
static std::mutex g_h3w_mu;

static H3WScope *h3w_scope_of(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_h3w_mu);
    for (int i = 0; i < H3W_MAX_SCOPES; ++i)
        if (g_h3w[i].active && g_h3w[i].stream == st) return &g_h3w[i];
    return nullptr;
}

extern "C" int64_t isc_splitk_workspace_bytes(int64_t M, int64_t N) {
    if (M <= 0 || N <= 0) return 0;
    const int64_t b = 16 * M * N * (int64_t)sizeof(float);            // plan_splitk: S <= 16 slabs of [M, N]
    return (b + 255) / 256 * 256;
}
extern "C" int64_t isc_h3_weights_workspace_bytes(int64_t weight_elements, int with_transposes) {
    if (weight_elements <= 0) return 0;
    const int64_t one = weight_elements * 2 * (int64_t)sizeof(uint16_t);     // hi + lo f16 plane per value
    const int64_t b = one * (with_transposes ? 2 : 1);
    return (b + 1048575) / 1048576 * 1048576;
}

extern "C" int isc_h3_weights_begin(void *buf, long long bytes, void *stream) {
    if (!buf || bytes <= 0) return ISC_E_NULL;
    if ((uintptr_t)buf & 255) return ISC_E_ALIGN;
    std::lock_guard<std::mutex> lk(g_h3w_mu);
    H3WScope *slot = nullptr;
    for (int i = 0; i < H3W_MAX_SCOPES; ++i)                     // the stream's own slot (active or suspended)
        if (g_h3w[i].buf && g_h3w[i].stream == (hipStream_t)stream) slot = &g_h3w[i];
    for (int i = 0; !slot && i < H3W_MAX_SCOPES; ++i)            // a free one
        if (!g_h3w[i].buf) slot = &g_h3w[i];
    // No take-over of another stream's SUSPENDED slot: its owner may be a captured graph that still reads those planes
    // (round 4: a training graph computed a loss from planes one update old after such a take-over).  More live scopes
    // than slots: this caller runs without one (every launch splits its weights into the workspace).
    if (!slot) return ISC_E_WORKSPACE;
    slot->stream = (hipStream_t)stream;
    slot->buf = static_cast<char *>(buf); slot->bytes = (size_t)bytes; slot->used = 0; slot->n = 0;
    slot->active = true;
    return ISC_OK;
}
extern "C" int isc_h3_weights_end(void *stream) {
    std::lock_guard<std::mutex> lk(g_h3w_mu);
    for (int i = 0; i < H3W_MAX_SCOPES; ++i)
        if (g_h3w[i].buf && g_h3w[i].stream == (hipStream_t)stream) {
            g_h3w[i].active = false; g_h3w[i].n = 0; g_h3w[i].used = 0; g_h3w[i].buf = nullptr;
        }
    return ISC_OK;
}
extern "C" int isc_h3_weights_suspend(void *stream) {
    std::lock_guard<std::mutex> lk(g_h3w_mu);
    for (int i = 0; i < H3W_MAX_SCOPES; ++i)
        if (g_h3w[i].active && g_h3w[i].stream == (hipStream_t)stream) g_h3w[i].active = false;
    return ISC_OK;
}
extern "C" int isc_h3_weights_resume(void *buf, void *stream) {
    if (!buf) return ISC_E_NULL;
    std::lock_guard<std::mutex> lk(g_h3w_mu);
    for (int i = 0; i < H3W_MAX_SCOPES; ++i)
        if (!g_h3w[i].active && g_h3w[i].buf == static_cast<char *>(buf) && g_h3w[i].stream == (hipStream_t)stream) {
            g_h3w[i].active = true;
            return ISC_OK;
        }
    return ISC_E_STATE;
}

// Re-split every weight the stream's scope (active or suspended) holds planes of, from the weights' CURRENT values, into
// the same planes: what an optimiser step calls after it has written the weights in place, so that the next forward /
// backward sweeps resume the scope instead of rebuilding it plane by plane (39 one-job split launches per XE iteration at
// B = 128 became ceil(entries / H3_MAX_JOBS) launches).  Enqueued on `stream`, i.e. ordered after whatever wrote the
// weights there; the caller orders other streams.
extern "C" int isc_h3_weights_refresh(void *stream) {
    H3WScope *sc = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_h3w_mu);
        for (int i = 0; i < H3W_MAX_SCOPES; ++i)
            if (g_h3w[i].buf && g_h3w[i].stream == (hipStream_t)stream) sc = &g_h3w[i];
    }
    if (!sc) return ISC_E_STATE;
    int i = 0;
    while (i < sc->n) {
        SplitLaunch S = {};
        int blocks = 0;
        for (; i < sc->n && S.njobs < H3_MAX_JOBS; ++i) {
            const H3WEntry &e = sc->e[i];
            SplitJob &J = S.j[S.njobs++];
            int k0 = 0;
            for (int sg = 0; sg < e.nseg; ++sg) {
                J.src[sg] = e.W[sg]; J.ld[sg] = e.ldw[sg]; J.kstart[sg] = k0;
                k0 += e.K[sg];
            }
            J.nseg = e.nseg; J.rows = e.rows; J.Kp = k0; J.first_block = blocks;
            J.hi = const_cast<_Float16 *>(e.hi); J.lo = const_cast<_Float16 *>(e.lo);
            J.transposed = e.transposed;
            blocks += e.transposed ? ((e.rows + 63) / 64) * (k0 >> 5) : (int)(((long long)e.rows * (k0 >> 3) + 255) / 256);
        }
        hipLaunchKernelGGL(h3_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S);
        ISC_LAUNCH_CHECK();
    }
    return ISC_OK;
}

static const H3WEntry *h3w_find(const H3WScope *sc, const DevProb &p, int transposed = 0) {
    if (!sc) return nullptr;
    for (int i = 0; i < sc->n; ++i) {
        const H3WEntry &e = sc->e[i];
        bool eq = e.nseg == p.nseg && e.rows == p.N && e.transposed == transposed;
        for (int s = 0; eq && s < p.nseg; ++s)
            eq = e.W[s] == p.seg[s].W && e.ldw[s] == p.seg[s].ldw && e.K[s] == p.seg[s].K;
        if (eq) return &e;
    }
    return nullptr;
}

template <int EPI>
static int launch_h3x(const DevLaunch &L, hipStream_t st) {
    constexpr size_t lds = 3 * (2 * 256 * 64 + 2 * 128 * 64);             // 147456: one workgroup per CU
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_h3x_kernel<EPI, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_h3x_kernel<EPI, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set.store(true);
    }
    if (h3_any_f32(L)) hipLaunchKernelGGL((gemm_h3x_kernel<EPI, true>), dim3(L.total_tiles), dim3(512), lds, st, L);
    else hipLaunchKernelGGL((gemm_h3x_kernel<EPI, false>), dim3(L.total_tiles), dim3(512), lds, st, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// 256-row tiles when they fill the chip, else the 128-row kernel.  The vocabulary projection stays on the 128-row
// kernel: its epilogue (per-row softmax statistics, ~3k VALU instructions per wave) is as long as its 16-chunk main
// loop, and only a second, out-of-phase workgroup on the CU overlaps the two (174 us on the 256-row tile vs 162 us;
// a persistent form of the 256-row kernel that put the next tile's first two chunks in flight before the epilogue
// measured 173 us: what is exposed is the epilogue's own VALU time, 8-10 us per tile, not the workgroup turnover).
template <int EPI>
static int launch_h3_big(DevLaunch &L, hipStream_t st) {
    long long t256 = 0;
    for (int i = 0; i < L.nprob; ++i) t256 += (long long)((L.p[i].M + 255) / 256) * ((L.p[i].N + 127) / 128);
    if (EPI != EPI_VOCAB && t256 >= 224) {        // (round 3: 160 - the three h-projections on this kernel - measured the same)
        ++g_h3x_launches;
        finish_tiling(L, 3);
        return launch_h3x<EPI>(L, st);
    }
    finish_tiling(L, 4);
    return launch_h3<EPI>(L, st);
}

static int launch_h3m(const DevLaunch &L, hipStream_t st) {
    constexpr size_t lds = H3M_NBUF * (2 * 64 * 64 + 2 * 128 * 64);      // 4 x 24 KB = 98304: one workgroup per CU
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_h3m_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_h3m_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set.store(true);
    }
    if (h3_any_f32(L)) hipLaunchKernelGGL(gemm_h3m_kernel<true>, dim3(L.total_tiles), dim3(256), lds, st, L);
    else hipLaunchKernelGGL(gemm_h3m_kernel<false>, dim3(L.total_tiles), dim3(256), lds, st, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// Operand-split jobs of one launch: add() lays the planes of an operand out in the workspace and queues its job.
struct H3Planner {
    SplitLaunch S = {};
    char *at;
    int blocks = 0;
    H3WScope *scope;                                   // the launch stream's weights scope, or null
    explicit H3Planner(float *ws, H3WScope *sc = nullptr) : at(reinterpret_cast<char *>(ws)), scope(sc) {}
    // planes of segments [s0, s1) of an operand, K-packed
    void add(const DevProb &p, bool is_w, int rows, const _Float16 *&hi, const _Float16 *&lo, int s0 = 0, int s1 = -1,
             int transposed = 0) {
        if (s1 < 0) s1 = p.nseg;
        SplitJob &J = S.j[S.njobs++];
        J.transposed = transposed;
        int k0 = 0;
        for (int s = s0; s < s1; ++s) {
            J.src[s - s0] = is_w ? p.seg[s].W : p.seg[s].A;
            J.ld[s - s0] = is_w ? p.seg[s].ldw : p.seg[s].lda;
            J.kstart[s - s0] = k0;
            k0 += p.seg[s].K;
        }
        J.nseg = s1 - s0; J.rows = rows; J.Kp = k0; J.first_block = blocks;
        const size_t bytes = (((size_t)rows * k0 * 4) + 255) & ~(size_t)255;      // hi and lo interleaved
        J.hi = reinterpret_cast<_Float16 *>(at); at += bytes;
        J.lo = J.hi + 32;
        hi = J.hi; lo = J.lo;
        blocks += transposed ? ((rows + 63) / 64) * (k0 >> 5) : (int)(((long long)rows * (k0 >> 3) + 255) / 256);
    }
    // activation operand: the caller's per-tensor planes where given; a segment without planes is read as fp32 rows
    // and split in registers after the fragment read (ap.hi == nullptr) - no split launch, no plane copy in memory
    void add_a(DevProb &p) {
        p.nap = p.nseg;
        for (int s = 0; s < p.nseg; ++s) {
            const int K = p.seg[s].K;
            const bool planes = p.seg[s].A_hi && p.seg[s].A_lo;
            p.ap[s] = DevASeg{planes ? p.seg[s].A_hi : nullptr, planes ? p.seg[s].A_lo : nullptr, 2 * K, K};
        }
    }
    // weight operand: cached planes if the caller opened a weights scope, else planes in the workspace
    void add_w(const DevProb &p, const _Float16 *&hi, const _Float16 *&lo, int transposed = 0) {
        if (const H3WEntry *e = h3w_find(scope, p, transposed)) { hi = e->hi; lo = e->lo; return; }
        const size_t bytes = (((size_t)p.N * h3_kp(p) * 4) + 255) & ~(size_t)255;
        if (scope && scope->n < H3W_MAX_ENTRIES && scope->used + bytes <= scope->bytes) {
            char *keep = at;
            at = scope->buf + scope->used;
            add(p, true, p.N, hi, lo, 0, -1, transposed);
            at = keep;
            scope->used += bytes;
            H3WEntry &e = scope->e[scope->n++];
            e.nseg = p.nseg; e.rows = p.N; e.hi = hi; e.lo = lo; e.transposed = transposed;
            for (int s = 0; s < p.nseg; ++s) { e.W[s] = p.seg[s].W; e.ldw[s] = p.seg[s].ldw; e.K[s] = p.seg[s].K; }
            return;
        }
        add(p, true, p.N, hi, lo, 0, -1, transposed);
    }
    int launch(hipStream_t st) {
        if (!S.njobs) return ISC_OK;
        hipLaunchKernelGGL(h3_split_kernel, dim3(blocks), dim3(256), 0, st, S);
        ISC_LAUNCH_CHECK();
        return ISC_OK;
    }
};

// Plans the planes of every problem inside the caller's workspace, launches the operand split and the GEMM.
// Returns 1 when the launch went out on this path (rc = its status), 0 when the path does not apply.
template <int EPI>
static int try_h3(DevLaunch &L, float *ws, long long ws_floats, hipStream_t st, int &rc, int transposed = 0) {
    const int h3_mode = g_h3_mode.load();
    if (h3_mode == 0 || g_tile_override.load() >= 0 || !ws || ((uintptr_t)ws & 255)) return 0;
    long long tiles = 0, need = 0;
    for (int i = 0; i < L.nprob; ++i) {
        const DevProb &p = L.p[i];
        if (p.ksplit > 1) return 0;
        tiles += (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        const long long Kp = h3_kp(p);
        if (Kp > (1 << 20)) return 0;
        need += (long long)p.N * Kp + 1024;               // floats: the weight planes (2 x 2 bytes per element) when the
                                                          // stream has no weights scope; activations are never copied
    }
    // linear launches that cannot give every CU a 128-row tile: the 64-row tile, unless it then needs more rounds of
    // the chip than it saves in bytes per workgroup (rounds x (rows + 128 columns) of operand lines per workgroup:
    // [4608 x 512] K=2048 is 288 64-row tiles = two rounds, 97 us, against one round of 144 128-row tiles)
    long long tiles64 = 0;
    for (int i = 0; i < L.nprob; ++i) tiles64 += (long long)((L.p[i].M + 63) / 64) * ((L.p[i].N + 127) / 128);
    const bool half_tile = EPI == EPI_LINEAR && tiles < 256 &&
                           ((tiles64 + 255) / 256) * 192 < ((tiles + 255) / 256) * 256;
    H3WScope *scope = h3w_scope_of(st);
    // inside a weights scope the planes are already there and the skinny kernel has taken what it does better
    // (try_h3s ran first): whatever is left with a few tiles is still faster here than on the fp32 tiles
    const long long min_tiles = scope && h3_mode == 1 ? H3_MIN_TILES_SCOPE
                                                      : (EPI == EPI_LINEAR && tiles < 256 ? H3_MIN_TILES / 2 : H3_MIN_TILES);
    if (h3_mode != 2 && tiles < min_tiles) return 0;
    if (need > ws_floats) return 0;
    {
        int jobs = 0;
        for (int i = 0; i < L.nprob; ++i) jobs += 1;
        if (jobs > H3_MAX_JOBS) return 0;
    }
    H3Planner pl(ws, scope);
    for (int i = 0; i < L.nprob; ++i) {
        DevProb &p = L.p[i];
        p.Kp = h3_kp(p);
        pl.add_a(p);
        pl.add_w(p, p.Wh, p.Wl, transposed);
    }
    rc = pl.launch(st);
    if (rc) return 1;
    if (half_tile) {
        finish_tiling(L, 1);
        rc = launch_h3m(L, st);
    } else {
        // a long contraction on few 128 x 128 tiles (one problem, linear epilogue): S slices of K so that the launch fills
        // the chip's 512 workgroup slots, raw partial tiles to slabs behind the planes in the workspace, then the reduce
        int S = 1;
        // (backward dX contractions only - `transposed` - unless forced: a forward launch keeps ONE summation order at every
        // batch size, so that a ReLU's sign at a pre-activation of ~0 does not depend on how many rows share the launch)
        if (EPI == EPI_LINEAR && L.nprob == 1 && (L.p[0].N & 3) == 0 && (g_h3_ksplit.load() == 2 || (transposed && g_h3_ksplit.load()))) {
            DevProb &p = L.p[0];
            const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 127) / 128);
            const int nblk = p.Kp / 32;
            if (t256 < 224 && tiles <= 200 && nblk >= 64) {
                S = (int)(512 / tiles);
                if (S > 8) S = 8;
                while (S > 1 && nblk / S < 24) --S;
                const long long used = (need + 255) & ~255LL, slab = (long long)p.M * p.N;
                while (S > 1 && used + S * slab > ws_floats) --S;
                if (S > 1) {
                    p.ksplit = S;
                    p.slab = ws + used;
                    p.slab_stride = slab;
                }
            }
        }
        rc = launch_h3_big<EPI>(L, st);
        if (!rc && S > 1) rc = launch_splitk_linear_reduce(L, st);
        if (S > 1) L.p[0].ksplit = 0;
    }
    ++g_h3_launches;
    return 1;
}

// ---- skinny split-f16 path (gemm_h3s_kernel): few-row launches inside a weights scope ----
static std::atomic<long long> g_h3s_launches{0};
extern "C" long long isc_h3s_launches(void) { return g_h3s_launches.load(); }
#define H3S_MAX_ROWS 2048
#define H3S_MAX_ROWS_NN 4096
#define H3S_EIGHT_WAVES_MIN_K 1024
#define H3S_MAX_WGS_VOCAB 384

// (tests / A-B runs: 0 sends vocabulary launches of the skinny path back to gemm_h3s_kernel<EPI_VOCAB>)
static std::atomic<int> g_h3v_on{1};
extern "C" int isc_set_h3v(int on) { return g_h3v_on.exchange(on < 0 ? 0 : on); }

template <int EPI, int T, int NW = 4>
static int launch_h3s(const DevLaunch &L, hipStream_t st) {
    // waves x ring slots x (A image + W image): 128 KB for the K-split tiles, 64 KB for the vocabulary projection
    constexpr size_t lds = (size_t)NW * (EPI != EPI_VOCAB && T == 1 && NW == 4 ? 4 : 2) * 8192 * T;
    static std::atomic<bool> attr_set{false};
    if (lds > 65536 && !attr_set.load()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_h3s_kernel<EPI, T, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set.store(true);
    }
    hipLaunchKernelGGL((gemm_h3s_kernel<EPI, T, NW>), dim3(L.total_tiles), dim3(64 * NW), lds, st, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

// Tile choice of a skinny launch.  One workgroup per CU and every workgroup ingest-bound (~35-40 GB/s per CU), so a
// launch costs ~4 us + its rounds of the chip x the bytes a workgroup takes in: per k ~6.5 ns for the 32 x 32 tile,
// twice that for the 64 x 64 tile (four times the outputs), against ~30 ns for a 128 x 128 tile of the large kernels
// (kernel-trace medians of tools/skinny_bench.py in modes 3 / 4 / 2, M = 128 ... 2048; LSTM cell K = 1536: M = 512
// 43 / 24 / 41 us, M = 1024 79 / 45 / 53 us, M = 2048 149 / 88 / 61 us).  So: the tile with fewer weighted rounds, and
// the large kernels past ~4.6 of them.  The vocabulary tile (32 x 128, two workgroups per CU) crosses earlier:
// M = 128 24 vs 28 us, M = 256 38 vs 22 us.
// Returns the tile factor T (1 or 2), or 0 when the large kernels should take the launch (auto mode only).
#define H3S_WIDE_COST 2.0
#define H3S_MAX_COST 4.6
template <int EPI>
static int h3s_pick_tile(const DevLaunch &L, int mode) {
    long long w1 = 0, w2 = 0;
    bool wide_ok = EPI != EPI_VOCAB;
    for (int i = 0; i < L.nprob; ++i) {
        const DevProb &p = L.p[i];
        w1 += (long long)((p.M + 31) / 32) * ((p.N + (EPI == EPI_VOCAB ? 127 : 31)) / (EPI == EPI_VOCAB ? 128 : 32));
        w2 += (long long)((p.M + 63) / 64) * ((p.N + 63) / 64);
        if (EPI == EPI_LSTM && (p.H & 15)) wide_ok = false;
    }
    if (mode == 4) return wide_ok ? 2 : 1;
    if (mode == 3) return 1;
    // (with gemm_h3v_kernel up to 512 workgroups = one round at two per CU: M = 160 / 192 roll-outs 1.67 / 1.79 -> 1.63 / 1.78 ms)
    if (EPI == EPI_VOCAB) return w1 < (g_h3v_on.load() > 1 ? g_h3v_on.load() : g_h3v_on.load() == 1 ? 512 : H3S_MAX_WGS_VOCAB) ? 1 : 0;
    const double c1 = (double)((w1 + 255) / 256), c2 = wide_ok ? H3S_WIDE_COST * (double)((w2 + 255) / 256) : 1e30;
    if ((c1 < c2 ? c1 : c2) >= H3S_MAX_COST) return 0;
    return c2 < c1 ? 2 : 1;
}

static void h3s_tile_problem(DevProb &p, int bm, int bn, int &start, int ksplit = 1) {
    p.ksplit = ksplit;
    p.tiles_m = (p.M + bm - 1) / bm;
    p.tiles_n = (p.N + bn - 1) / bn;
    p.tile_start = start;
    // One slice of the column tiles per XCD (map_tile's grouped order) once there are several row tiles: the XCD's
    // workgroups then share a weight slice of 1/8 that its L2 keeps, instead of every XCD streaming most of W.
    p.m_fastest = p.tiles_m > 1 ? 1 : 0;
    p.grp_n = (p.tiles_n + 7) / 8;
    start += p.tiles_m * p.tiles_n * ksplit;
}

static int launch_splitk_linear_reduce(const DevLaunch &L, hipStream_t st);

// Cross-workgroup K split of a skinny linear launch: a long contraction on few tiles (the classifier's dX at B = 128:
// [2560 x 512] over K = 9984 is 320 wide tiles = 1.25 rounds of ~130 us each) packs the chip better in S slices -
// cost in k per workgroup-round, as h3s_pick_tile, plus ~1000 for the reduce launch.  S <= 8, >= 16 k-blocks per slice.
static int h3s_pick_ksplit(const DevLaunch &L, int T, long long slab_floats) {
    long long w = 0, mn = 0;
    int kp_min = 1 << 30, kp_max = 0;
    for (int i = 0; i < L.nprob; ++i) {
        const DevProb &p = L.p[i];
        if (p.N & 3) return 1;
        w += (long long)((p.M + 32 * T - 1) / (32 * T)) * ((p.N + 32 * T - 1) / (32 * T));
        mn += (long long)p.M * p.N;
        const int kp = h3_kp(p);
        if (kp < kp_min) kp_min = kp;
        if (kp > kp_max) kp_max = kp;
    }
    if (kp_min < 4096) return 1;
    int best = 1;
    long long best_cost = ((w + 255) / 256) * kp_max;
    for (int S = 2; S <= 8; ++S) {
        if (kp_min / S < 512 || S * mn > slab_floats) break;
        const long long cost = ((w * S + 255) / 256) * (kp_max / S) + 1000;
        if (cost < best_cost) { best_cost = cost; best = S; }
    }
    return best;
}

template <int EPI>
static int launch_h3s_t(const DevLaunch &L, int T, hipStream_t st) {
    if constexpr (EPI != EPI_VOCAB) {
        if (T == 2) return launch_h3s<EPI, 2>(L, st);
        int kp_min = 1 << 30;
        for (int i = 0; i < L.nprob; ++i) {
            const int kq = L.p[i].Kp / (L.p[i].ksplit > 1 ? L.p[i].ksplit : 1);
            if (kq < kp_min) kp_min = kq;
        }
        // from 32 k-blocks on, eight waves (four blocks or more each) beat four when activation segments arrive as
        // fp32 rows (training: the in-register split is shared by twice the waves; kernel-trace sums of an XE
        // iteration at B = 128: LSTM cells -9 %, linears -3 %); with every operand on planes (the decode loop) four
        // waves with four-slot rings are the faster form (LSTM cell 9.5 vs 10.3 us)
        if (kp_min >= H3S_EIGHT_WAVES_MIN_K && h3_any_f32(L)) return launch_h3s<EPI, 1, 8>(L, st);
    }
    if constexpr (EPI == EPI_VOCAB) {
        // one activation segment (the classifier over h_lang): workgroup-shared k-block stages (gemm_h3v_kernel)
        bool v_ok = g_h3v_on.load() != 0;
        for (int i = 0; i < L.nprob; ++i)
            if (L.p[i].nap != 1 || L.p[i].ksplit > 1) v_ok = false;
        if (v_ok) {
            const size_t lds = H3V_STAGES * H3V_STAGE_BYTES;
            static std::atomic<bool> attr_set{false};
            if (lds > 65536 && !attr_set.load()) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_h3v_kernel<true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e == hipSuccess)
                    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_h3v_kernel<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return (int)e;
                attr_set.store(true);
            }
            if (h3_any_f32(L)) hipLaunchKernelGGL((gemm_h3v_kernel<true>), dim3(L.total_tiles), dim3(256), lds, st, L);
            else hipLaunchKernelGGL((gemm_h3v_kernel<false>), dim3(L.total_tiles), dim3(256), lds, st, L);
            ISC_LAUNCH_CHECK();
            return ISC_OK;
        }
        // one workgroup per CU at most (M <= 96 at V = 10 000; beam rows): eight waves, two per column block
        if (L.total_tiles <= 256) return launch_h3s<EPI, 1, 8>(L, st);
    }
    return launch_h3s<EPI, 1>(L, st);
}

// Returns 1 when the launch went out on the skinny path (rc = its status), 0 when it does not apply: mode off / tile
// override, no weights scope on this stream in auto mode (the weight planes would have to be rebuilt per launch),
// more rows than H3S_MAX_ROWS, or a launch the large split-f16 kernels do better (h3s_pick_tile).
// Modes 3 / 4 force the 32 x 32 / 64 x 64 tile (tests): without a scope the weight planes then go to the workspace.
template <int EPI>
static int try_h3s(DevLaunch &L, float *ws, long long ws_floats, hipStream_t st, int &rc, int transposed = 0) {
    const int mode = g_h3_mode.load();
    if ((mode != 1 && mode != 3 && mode != 4) || g_tile_override.load() >= 0) return 0;
    H3WScope *sc = h3w_scope_of(st);
    if (!sc && mode == 1) return 0;
    long long need = 0;
    for (int i = 0; i < L.nprob; ++i) {
        const DevProb &p = L.p[i];
        // (dX contractions over a vocabulary-sized K - long K, few tiles - also run here above H3S_MAX_ROWS, K-split)
        if (p.M > (transposed && h3_kp(p) >= 4096 ? H3S_MAX_ROWS_NN : H3S_MAX_ROWS)) return 0;
        if (EPI == EPI_LSTM && (p.H & 7)) return 0;
        for (int sg = 0; sg < p.nseg; ++sg)
            if (p.seg[sg].K & 31) return 0;                 // (the backward entry point accepts other K)
        const long long Kp = h3_kp(p);
        if (Kp > (1 << 20)) return 0;
        need += (long long)p.N * Kp + 256;
    }
    const int T = h3s_pick_tile<EPI>(L, mode);
    if (!T) return 0;                                     // the large kernels' launch
    if (!ws || ((uintptr_t)ws & 255) || need > ws_floats) return 0;   // planes that do not fit the scope go here
    if (L.nprob > H3_MAX_JOBS) return 0;
    int S = 1;
    if constexpr (EPI == EPI_LINEAR) S = h3s_pick_ksplit(L, T, ws_floats - need - 64);
    float *slab = ws + ((ws_floats - 64) & ~63LL);        // slabs from the end of the workspace, planes from its start
    H3Planner pl(ws, sc);
    int start = 0;
    for (int i = 0; i < L.nprob; ++i) {
        DevProb &p = L.p[i];
        p.Kp = h3_kp(p);
        p.nap = p.nseg;
        for (int s = 0; s < p.nseg; ++s) {
            const bool planes = p.seg[s].A_hi && p.seg[s].A_lo;
            p.ap[s] = DevASeg{planes ? p.seg[s].A_hi : nullptr, planes ? p.seg[s].A_lo : nullptr, 2 * p.seg[s].K,
                              p.seg[s].K};
        }
        pl.add_w(p, p.Wh, p.Wl, transposed);
        h3s_tile_problem(p, 32 * T, EPI == EPI_VOCAB ? 128 : 32 * T, start, S);
        if (S > 1) {
            p.slab_stride = (long long)p.M * p.N;
            slab -= S * p.slab_stride;
            p.slab = slab;
        }
    }
    L.total_tiles = start;
    rc = pl.launch(st);                                   // weight planes not yet in the scope (first use only)
    if (rc) return 1;
    rc = launch_h3s_t<EPI>(L, T, st);
    ++g_h3s_launches;
    if (!rc && S > 1) rc = launch_splitk_linear_reduce(L, st);
    return 1;
}

// dW = dY^T X (ISC_LAYOUT_TN: both operands are [K_s, .] with the contraction over their ROWS) on the split-f16
// kernels: planes of dY^T [M, Kp] and X^T [N, Kp] are built by the transposing split into the workspace (both operands
// are activations: new every call, never cached), then the NT kernels run as for any linear problem - the large tiles,
// the 64-row tile or the skinny tile by size.  A single-segment problem whose planes exceed the workspace goes through it
// in K chunks that accumulate into C.  Problems of a launch are handled one by one; returns the bit mask of the problems
// that went out on this path (the caller runs the others - a segment with K % 32 != 0, a small contraction - on the
// fp32 tiles); rc != 0 reports a launch error.
static unsigned try_h3_tn(DevLaunch &L, float *ws, long long ws_floats, hipStream_t st, int &rc) {
    const int mode = g_h3_mode.load();
    if (mode == 0 || g_tile_override.load() >= 0 || !ws || ((uintptr_t)ws & 255)) return 0;
    long long chunk[3];
    unsigned take = 0;
    for (int i = 0; i < L.nprob; ++i) {
        const DevProb &p = L.p[i];
        long long Kp = 0;
        bool ok = true;
        for (int s = 0; s < p.nseg; ++s) {
            ok = ok && (p.seg[s].K & 31) == 0;
            Kp += p.seg[s].K;
        }
        if (2.0 * p.M * p.N * (double)Kp < 2.5e8 && mode != 2) ok = false;    // small: the fp32 tiles' launch is as quick
        const long long per_k = (long long)p.M + p.N;
        chunk[i] = Kp;
        if (ok && per_k * Kp + 1024 > ws_floats) {
            // planes larger than the workspace: K chunks that accumulate into C - a multi-segment problem segment by
            // segment (the [fc | label] block of the att-LSTM's dW at B >= 512 is two segments of T B rows: 22-45 GFLOP
            // that used to stay on the fp32 tiles, 0.2-0.4 ms per sweep)
            chunk[i] = ((ws_floats - 1024) / per_k) & ~31LL;
            if (chunk[i] < 512) ok = false;
        }
        if (ok) take |= 1u << i;
    }
    // Several dW contractions from ONE dY (the weight gradients of the layers that consumed the same pre-activation
    // gradient: lang-LSTM's three input blocks from dG2, att-LSTM's from dG1): the transposing split of dY happens once,
    // the problems share one launch of the skinny tiles - 3 x (split + GEMM) -> 1 + 1 launches, and dY^T's planes
    // (the larger operand: [2048, T B] against [512, T B]) are built once instead of three times.
    if (L.nprob >= 2 && take == (1u << L.nprob) - 1u) {
        bool same = true;
        long long n_sum = 0;
        const DevProb &p0 = L.p[0];
        for (int i = 0; i < L.nprob; ++i) {
            const DevProb &p = L.p[i];
            same = same && p.nseg == 1 && p.seg[0].A == p0.seg[0].A && p.seg[0].lda == p0.seg[0].lda &&
                   p.seg[0].K == p0.seg[0].K && p.M == p0.M &&
                   (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) < H3_MIN_TILES / 2;
            n_sum += p.N;
        }
        const long long K = p0.nseg == 1 ? p0.seg[0].K : 0;
        // planes of dY^T and of every X^T for kc rows at a time: the whole K when it fits the workspace, else equal chunks
        // (multiples of 32) that accumulate into C - the B = 512 / 1024 iterations contract over 10 240 / 22 080 rows
        const long long per_k = (long long)p0.M + n_sum;
        const long long kc_max = ((ws_floats - 1024LL * (L.nprob + 1)) / (per_k > 0 ? per_k : 1)) & ~31LL;
        if (same && K > 0 && kc_max >= 2048) {
            const long long nchunk = (K + kc_max - 1) / kc_max;
            const long long kc0 = ((K + nchunk - 1) / nchunk + 31) & ~31LL;
            for (long long k0 = 0; k0 < K; k0 += kc0) {
                const long long kc = K - k0 < kc0 ? K - k0 : kc0;
                DevLaunch Lc = {};
                Lc.nprob = L.nprob;
                H3Planner pl(ws, nullptr);
                const _Float16 *ah = nullptr, *al = nullptr;
                DevProb a0 = p0;
                a0.seg[0].A = p0.seg[0].A + k0 * p0.seg[0].lda;
                a0.seg[0].K = (int)kc;
                pl.add(a0, false, a0.M, ah, al, 0, -1, 1);
                for (int i = 0; i < L.nprob; ++i) {
                    DevProb &q = Lc.p[i];
                    q = L.p[i];
                    q.seg[0].A = L.p[i].seg[0].A + k0 * L.p[i].seg[0].lda;
                    q.seg[0].W = L.p[i].seg[0].W + k0 * L.p[i].seg[0].ldw;
                    q.seg[0].K = (int)kc;
                    if (k0 > 0) { q.accumulate = 1; q.bias0 = q.bias1 = q.bias2 = nullptr; }
                    pl.add(q, true, q.N, q.Wh, q.Wl, 0, -1, 1);
                    q.nap = 1;
                    q.ap[0] = DevASeg{ah, al, 2 * (int)kc, (int)kc};
                    q.nseg = 1;
                    q.Kp = (int)kc;
                    q.ksplit = 1;
                }
                rc = pl.launch(st);
                if (rc) return take;
                int T = h3s_pick_tile<EPI_LINEAR>(Lc, mode == 2 ? 1 : mode), start = 0;
                if (!T) T = 2;
                for (int i = 0; i < Lc.nprob; ++i) h3s_tile_problem(Lc.p[i], 32 * T, 32 * T, start);
                Lc.total_tiles = start;
                rc = launch_h3s_t<EPI_LINEAR>(Lc, T, st);
                if (rc) return take;
                ++g_h3_launches;
            }
            return take;
        }
    }
    for (int i = 0; i < L.nprob; ++i) {
        if (!(take & (1u << i))) continue;
        const DevProb &pfull = L.p[i];
        long long Kfull = 0;
        for (int s = 0; s < pfull.nseg; ++s) Kfull += pfull.seg[s].K;
        const bool by_seg = pfull.nseg > 1 && chunk[i] < Kfull;      // does not fit: one single-segment pass per segment
        const int npass = by_seg ? pfull.nseg : 1;
        for (int pass = 0; pass < npass; ++pass) {
        DevProb p0 = pfull;
        if (by_seg) {
            p0.nseg = 1;
            p0.seg[0] = pfull.seg[pass];
            if (pass > 0) { p0.accumulate = 1; p0.bias0 = p0.bias1 = p0.bias2 = nullptr; }
        }
        long long Kp = 0;
        for (int s = 0; s < p0.nseg; ++s) Kp += p0.seg[s].K;
        for (long long k0 = 0; k0 < Kp; k0 += chunk[i]) {
            const long long kc = Kp - k0 < chunk[i] ? Kp - k0 : chunk[i];
            DevLaunch L1 = {};
            L1.nprob = 1;
            DevProb &q = L1.p[0];
            q = p0;
            if (p0.nseg == 1) {                                   // (a multi-segment problem that fits is one launch)
                q.seg[0].A = p0.seg[0].A + k0 * p0.seg[0].lda;
                q.seg[0].W = p0.seg[0].W + k0 * p0.seg[0].ldw;
                q.seg[0].K = (int)kc;
            }
            q.Kp = (int)kc;
            if (k0 > 0) { q.accumulate = 1; q.bias0 = q.bias1 = q.bias2 = nullptr; }
            H3Planner pl(ws, nullptr);
            const _Float16 *ah, *al;
            pl.add(q, false, q.M, ah, al, 0, -1, 1);
            pl.add(q, true, q.N, q.Wh, q.Wl, 0, -1, 1);
            q.nap = 1;
            q.ap[0] = DevASeg{ah, al, 2 * (int)kc, (int)kc};
            q.nseg = 1;                                           // the kernels see one packed K-segment
            q.seg[0].K = (int)kc;
            q.ksplit = 1;
            rc = pl.launch(st);
            if (rc) return take;
            const long long tiles = (long long)((q.M + 127) / 128) * ((q.N + 127) / 128);
            if (tiles >= H3_MIN_TILES) {
                rc = launch_h3_big<EPI_LINEAR>(L1, st);
            } else if (tiles >= H3_MIN_TILES / 2) {
                finish_tiling(L1, 1);
                rc = launch_h3m(L1, st);
            } else {
                int T = h3s_pick_tile<EPI_LINEAR>(L1, mode == 2 ? 1 : mode), start = 0;
                if (!T) T = 2;
                h3s_tile_problem(q, 32 * T, 32 * T, start);
                L1.total_tiles = start;
                rc = launch_h3s_t<EPI_LINEAR>(L1, T, st);
            }
            if (rc) return take;
            ++g_h3_launches;
        }
        }
    }
    return take;
}

// Split-K plan for launches with too few tiles to occupy the chip (small M): every workgroup would
// otherwise walk the whole contraction serially (~0.8 us per 32-deep chunk).  Returns the split count
// (1 = no split) and carves one [S, M, N] slab region per problem out of the caller's workspace.
static int plan_splitk(DevLaunch &L, float *ws, long long ws_floats) {
    if (!ws || ws_floats <= 0) return 1;
    long long blocks = 0, need1 = 0;
    int min_chunks = 1 << 30;
    for (int i = 0; i < L.nprob; ++i) {
        const DevProb &p = L.p[i];
        if (p.N & 3) return 1;
        blocks += (long long)((p.M + 31) / 32) * ((p.N + 127) / 128);
        need1 += (long long)p.M * p.N;
        int ch = 0;
        for (int s = 0; s < p.nseg; ++s) ch += (p.seg[s].K + BK - 1) / BK;
        if (ch < min_chunks) min_chunks = ch;
    }
    if (blocks >= 384) return 1;
    long long S = 768 / blocks;
    if (S > 16) S = 16;
    if (S > min_chunks / 4) S = min_chunks / 4;      // keep >= 4 chunks per split
    if (S > ws_floats / need1) S = ws_floats / need1;
    if (S < 2) return 1;
    float *at = ws;
    for (int i = 0; i < L.nprob; ++i) {
        DevProb &p = L.p[i];
        p.ksplit = (int)S;
        p.slab = at;
        p.slab_stride = (long long)p.M * p.N;
        at += S * p.slab_stride;
    }
    return (int)S;
}

static int launch_splitk_linear_reduce(const DevLaunch &L, hipStream_t st) {
    long long mx = 0;
    for (int i = 0; i < L.nprob; ++i) {
        const long long n4 = (long long)L.p[i].M * (L.p[i].N >> 2);
        if (n4 > mx) mx = n4;
    }
    hipLaunchKernelGGL(splitk_linear_kernel, dim3((unsigned)((mx + 255) / 256), L.nprob), dim3(256), 0, st, L);
    ISC_LAUNCH_CHECK();
    return ISC_OK;
}

extern "C" int isc_linear_fwd(const isc_linear_problem *pr, int n_prob, void *stream) {
    if (!pr) return ISC_E_NULL;
    if (n_prob < 1 || n_prob > 3) return ISC_E_SHAPE;
    DevLaunch L = {};
    L.nprob = n_prob;
    L.gate = isc_stream_gate_(stream);
    for (int i = 0; i < n_prob; ++i) {
        const isc_linear_problem &q = pr[i];
        int rc = check_segs(q.seg, q.nseg);
        if (rc) return rc;
        if (!q.C) return ISC_E_NULL;
        if (q.M <= 0 || q.N <= 0) return ISC_E_SHAPE;
        if (q.keep_mask && q.ldc != q.N) return ISC_E_SHAPE;
        DevProb &d = L.p[i];
        copy_segs(d, q.seg, q.nseg);
        d.M = q.M; d.N = q.N; d.relu = q.relu;
        d.bias0 = q.bias0; d.bias1 = q.bias1; d.bias2 = q.bias2;
        d.mask = q.keep_mask; d.mask_scale = q.mask_scale;
        d.ldc = q.ldc; d.C = q.C; d.C_pre = q.C_pre; d.accumulate = q.accumulate;
    }
    int rc = ISC_OK;
    if (try_gemv<EPI_LINEAR>(L, (hipStream_t)stream, rc)) return rc;
    if (try_h3s<EPI_LINEAR>(L, pr[0].splitk_ws, pr[0].splitk_ws_floats, (hipStream_t)stream, rc)) return rc;
    const int S = plan_splitk(L, pr[0].splitk_ws, pr[0].splitk_ws_floats);
    if (S == 1 && try_h3<EPI_LINEAR>(L, pr[0].splitk_ws, pr[0].splitk_ws_floats, (hipStream_t)stream, rc)) return rc;
    const int tile = S > 1 ? 2 : pick_tile(L, true);
    finish_tiling(L, tile);
    rc = launch_any<EPI_LINEAR, false, false>(L, tile, (hipStream_t)stream);
    if (rc || S == 1) return rc;
    return launch_splitk_linear_reduce(L, (hipStream_t)stream);
}

// Backward-pass contractions on the same kernel (include/insenticap_hip.h: isc_gemm_bwd).
extern "C" int isc_gemm_bwd(const isc_linear_problem *pr, int n_prob, int layout, void *stream) {
    if (!pr) return ISC_E_NULL;
    if (n_prob < 1 || n_prob > 3 || (layout != ISC_LAYOUT_NN && layout != ISC_LAYOUT_TN)) return ISC_E_SHAPE;
    // dW = sum over segments of A_s^T W_s where some segments have a row count the split-f16 kernels cannot take (K % 32
    // != 0: the once-per-caption block of the att-LSTM's dW has B rows, 80 in the seq2seq unroll) and others are large:
    // two launches - the K % 32 == 0 segments first (split-f16), the rest accumulating on the fp32 tiles - instead of
    // the whole sum on the fp32 tiles (3.5 GFLOP, 82 us per iteration)
    if (layout == ISC_LAYOUT_TN && n_prob == 1 && pr[0].nseg > 1 && pr[0].nseg <= ISC_MAX_SEG && g_h3_mode.load() != 0) {
        isc_linear_problem a = pr[0], b = pr[0];
        a.nseg = b.nseg = 0;
        double fa = 0;
        for (int s = 0; s < pr[0].nseg; ++s) {
            if (pr[0].seg[s].K > 0 && (pr[0].seg[s].K & 31) == 0) {
                a.seg[a.nseg++] = pr[0].seg[s];
                fa += 2.0 * pr[0].M * pr[0].N * (double)pr[0].seg[s].K;
            } else {
                b.seg[b.nseg++] = pr[0].seg[s];
            }
        }
        if (a.nseg > 0 && b.nseg > 0 && fa >= 2.5e8) {
            const int rc = isc_gemm_bwd(&a, 1, layout, stream);
            if (rc) return rc;
            b.accumulate = 1; b.bias0 = b.bias1 = b.bias2 = nullptr;
            return isc_gemm_bwd(&b, 1, layout, stream);
        }
    }
    DevLaunch L = {};
    L.nprob = n_prob;
    for (int i = 0; i < n_prob; ++i) {
        const isc_linear_problem &q = pr[i];
        if (q.nseg < 1 || q.nseg > ISC_MAX_SEG || !q.C) return q.C ? ISC_E_SHAPE : ISC_E_NULL;
        if (q.M <= 0 || q.N <= 0 || (q.N & 3) || (q.ldc & 3)) return ISC_E_SHAPE;
        if (layout == ISC_LAYOUT_TN && (q.M & 3)) return ISC_E_SHAPE;
        for (int s = 0; s < q.nseg; ++s) {
            const isc_seg &g = q.seg[s];
            if (!g.A || !g.W) return ISC_E_NULL;
            // NN with K % 4 != 0 (vocabulary-sized K): the caller pads A's rows to lda with zeros
            if (g.K <= 0) return ISC_E_SHAPE;
            if ((g.lda & 3) || (g.ldw & 3) || !isc_aligned16(g.A) || !isc_aligned16(g.W)) return ISC_E_ALIGN;
        }
        DevProb &d = L.p[i];
        copy_segs(d, q.seg, q.nseg);
        d.M = q.M; d.N = q.N; d.relu = 0;
        d.bias0 = q.bias0; d.bias1 = q.bias1; d.bias2 = q.bias2;
        d.ldc = q.ldc; d.C = q.C; d.accumulate = q.accumulate;
    }
    int rc_h3 = ISC_OK;
    bool k32 = true;
    for (int i = 0; i < n_prob; ++i)
        for (int s = 0; s < pr[i].nseg; ++s) k32 = k32 && (pr[i].seg[s].K & 31) == 0;
    // ... or outside one when the contraction is large (>= 5 GFLOP: the classifier's dX over a zero-padded copy of W_c that
    // autograd.py keeps out of the scope on purpose - B = 512: 105 GFLOP, 606 us on the fp32 tiles -, the prologue's
    // d att_e = d att_p W at B >= 256: 19 GFLOP, 283 us at B = 1024): planes of this call's W^T go to the workspace, one
    // split launch of at most 20 MB against 0.2-0.3 ms saved
    bool nn_big = g_h3_mode.load() == 1 && n_prob == 1;
    if (nn_big) {
        double fl = 0;
        for (int s = 0; s < pr[0].nseg; ++s) fl += 2.0 * pr[0].M * pr[0].N * (double)pr[0].seg[s].K;
        nn_big = fl >= 5.0e9;
    }
    if (layout == ISC_LAYOUT_NN && k32 && (h3w_scope_of((hipStream_t)stream) || g_h3_mode.load() >= 2 || nn_big)) {
        // inside a weights scope (the BPTT sweep): dX = dY W on planes of W^T built once per scope - the skinny tiles for
        // few rows, the large split-f16 kernels otherwise; dY is read as fp32 rows and split in registers
        if (try_h3s<EPI_LINEAR>(L, pr[0].splitk_ws, pr[0].splitk_ws_floats, (hipStream_t)stream, rc_h3, 1)) return rc_h3;
        if (try_h3<EPI_LINEAR>(L, pr[0].splitk_ws, pr[0].splitk_ws_floats, (hipStream_t)stream, rc_h3, 1)) return rc_h3;
    }
    if (layout == ISC_LAYOUT_TN) {
        const unsigned took = try_h3_tn(L, pr[0].splitk_ws, pr[0].splitk_ws_floats, (hipStream_t)stream, rc_h3);
        if (rc_h3) return rc_h3;
        if (took) {                                     // the rest of the launch stays on the fp32 tiles
            int n = 0;
            for (int i = 0; i < L.nprob; ++i)
                if (!(took & (1u << i))) { if (n != i) L.p[n] = L.p[i]; ++n; }
            if (n == 0) return ISC_OK;
            L.nprob = n;
        }
    }
    const int S = plan_splitk(L, pr[0].splitk_ws, pr[0].splitk_ws_floats);
    const int tile = S > 1 ? 2 : pick_tile(L, false, false);
    finish_tiling(L, tile);
    int rc = layout == ISC_LAYOUT_NN ? launch_any<EPI_LINEAR, false, true>(L, tile, (hipStream_t)stream)
                                     : launch_any<EPI_LINEAR, true, true>(L, tile, (hipStream_t)stream);
    if (rc || S == 1) return rc;
    return launch_splitk_linear_reduce(L, (hipStream_t)stream);
}

extern "C" int isc_lstm_fwd(const isc_lstm_problem *q, void *stream) {
    if (!q) return ISC_E_NULL;
    int rc = check_segs(q->seg, q->nseg);
    if (rc) return rc;
    if (!q->c_prev || !q->h_out || !q->c_out) return ISC_E_NULL;
    if ((!q->b_ih || !q->b_hh) && !q->pre) return ISC_E_NULL;       // biases may only be folded into `pre`
    if ((q->b_ih == nullptr) != (q->b_hh == nullptr)) return ISC_E_NULL;
    if (q->tab && !q->tab_ids) return ISC_E_NULL;
    if (q->M <= 0 || q->H <= 0 || (q->H % 32) != 0) return ISC_E_SHAPE;
    if (q->h_keep_mask && !q->hdrop_out) return ISC_E_NULL;
    DevLaunch L = {};
    L.nprob = 1;
    L.gate = isc_stream_gate_(stream);
    DevProb &d = L.p[0];
    copy_segs(d, q->seg, q->nseg);
    d.M = q->M; d.N = 4 * q->H; d.H = q->H;
    d.bias0 = q->b_ih; d.bias1 = q->b_hh;
    d.c_prev = q->c_prev; d.h_out = q->h_out; d.c_out = q->c_out; d.gates_out = q->gates_out;
    d.hmask = q->h_keep_mask; d.mask_scale = q->mask_scale; d.hdrop = q->hdrop_out;
    if ((q->h_hi == nullptr) != (q->h_lo == nullptr)) return ISC_E_NULL;
    d.h_hi = static_cast<_Float16 *>(q->h_hi); d.h_lo = static_cast<_Float16 *>(q->h_lo);
    d.pre = q->pre; d.tab = q->tab; d.tab_ids = q->tab_ids; d.tab_ids_stride = q->tab_ids_stride;
    if (try_gemv<EPI_LSTM>(L, (hipStream_t)stream, rc)) return rc;
    if (try_h3s<EPI_LSTM>(L, q->splitk_ws, q->splitk_ws_floats, (hipStream_t)stream, rc)) return rc;
    const int S = plan_splitk(L, q->splitk_ws, q->splitk_ws_floats);
    if (S > 1) {   // plain [M,4H] pre-activation slabs, then the cell update in the reduce kernel
        finish_tiling(L, 2);
        int rc2 = launch_any<EPI_LINEAR, false, false>(L, 2, (hipStream_t)stream);
        if (rc2) return rc2;
        const long long n = (long long)q->M * q->H;
        hipLaunchKernelGGL(splitk_lstm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, L);
        ISC_LAUNCH_CHECK();
        return ISC_OK;
    }
    // XL only for the bias-only cell: a hoisted `pre` term or an embedding-table gather adds 8-16 B of
    // epilogue reads per output, which the XL tile's lone workgroup per CU cannot overlap with MFMA work
    // (in the roll-out: att-LSTM 168 us on the 128x128 tile vs 178 us on XL; lang-LSTM 223 vs 206; with the
    // batched epilogue loads of lstm_cells the two are within 2 % of each other on either cell)
    if (try_h3<EPI_LSTM>(L, q->splitk_ws, q->splitk_ws_floats, (hipStream_t)stream, rc)) return rc;
    const int tile = pick_tile(L, !q->pre && !q->tab, true, true);
    finish_tiling(L, tile);
    return launch_any<EPI_LSTM, false, false>(L, tile, (hipStream_t)stream);
}

extern "C" int isc_vocab_fwd(const float *h, int ldh, const float *W, int ldw, const float *bias,
                             int M, int V, int K, float *logits, int64_t ld_logits,
                             float *part_max, float *part_sum, int32_t *part_idx,
                             const void *h_hi, const void *h_lo,
                             float *splitk_ws, int64_t splitk_ws_floats, void *stream) {
    if (!h || !W || !bias || !part_max || !part_sum || !part_idx) return ISC_E_NULL;
    isc_seg sg = {h, W, ldh, ldw, K, 0, h_hi, h_lo};
    int rc = check_segs(&sg, 1);
    if (rc) return rc;
    if (M <= 0 || V <= 0) return ISC_E_SHAPE;
    DevLaunch L = {};
    L.nprob = 1;
    L.gate = isc_stream_gate_(stream);
    DevProb &d = L.p[0];
    copy_segs(d, &sg, 1);
    d.M = M; d.N = V; d.bias0 = bias;
    d.C = logits; d.ld_logits = ld_logits;
    d.pmax = part_max; d.psum = part_sum; d.pidx = part_idx;
    d.ntile_total = (V + 127) / 128;
    // few rows (beam search, small batches): the 16-chunk contraction of a 32-row tile is a serial walk
    // of ~27 us; split it over K into raw [S,M,V] slabs and let the reduce kernel form the statistics
    if (try_gemv<EPI_VOCAB>(L, (hipStream_t)stream, rc)) return rc;
    if (try_h3s<EPI_VOCAB>(L, splitk_ws, splitk_ws_floats, (hipStream_t)stream, rc)) return rc;
    const int S = plan_splitk(L, splitk_ws, splitk_ws_floats);
    if (S > 1) {
        finish_tiling(L, 2);
        rc = launch_any<EPI_LINEAR, false, false>(L, 2, (hipStream_t)stream);
        if (rc) return rc;
        const long long waves = (long long)M * d.ntile_total;
        hipLaunchKernelGGL(splitk_vocab_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, L);
        ISC_LAUNCH_CHECK();
        return ISC_OK;
    }
    // The vocabulary projection runs on the 128x128 LDS-DMA tile: at [4096 x 10000 x 512] 121 TFLOP/s there, 115 on
    // the register-staged 128x128 tile and 108 on XL, whose lone workgroup per CU has nothing to hide the per-row
    // softmax statistics of the epilogue behind (3k VALU instructions per wave at the end of a 16-chunk tile).
    if (try_h3<EPI_VOCAB>(L, splitk_ws, splitk_ws_floats, (hipStream_t)stream, rc)) return rc;
    const int tile = pick_tile(L, false, true, true);
    finish_tiling(L, tile);
    return launch_any<EPI_VOCAB, false, false>(L, tile, (hipStream_t)stream);
}
