// Decode rows: the inference decode step (captioner.py:168-186) on at most 8 rows - the beam rows of one image
// (captioner.py:380-411), a roll-out of a handful of captions - for gfx950.
//
// With <= 8 rows every contraction of the step is a handful of matrix-vector products that stream their weights
// once (44.5 MB of fp32 per step at the reference sizes) and every launch is a few microseconds long: what a launch
// costs is its chain of dependent memory round trips, not its arithmetic.  The kernels here are built around that:
//   * weights go straight to registers, ALL of a wave's weight loads are issued before anything else that touches
//     vector memory (no LDS staging of activations in front of them, no barrier); the activation slices a wave
//     needs (<= 8 rows x 256 floats) are loaded behind them, also to registers;
//   * lanes: a weight row is read by ONE 16-lane group (256 contiguous bytes per instruction and row: whole
//     128-byte lines), four rows per instruction - the four gates of a hidden unit, or four vocabulary columns -
//     so that the cross-lane sum is four DPP adds inside a 16-lane row (no LDS crossbar, no 64-lane butterfly);
//     the K axis is cut into slices of <= 256 floats that different waves of the workgroup contract, partial sums
//     meet in LDS in fixed order (deterministic);
//   * dependent index chains (beam re-ordering: row -> source row of the state; token -> row of the token table)
//     sit in an extra wave of the workgroup that issues no weight loads, so the in-order vmcnt of the streaming
//     waves never waits on them; the streaming waves read the source-row indices through the scalar cache (lgkmcnt);
//   * the beam's state re-ordering is an index on the loads of h / c (isc_rows_ext.src_row): no gather launch;
//   * the classifier's epilogue leaves, per column tile, the log-softmax statistics AND the top-`beam` masked logits,
//     so that the top-k + candidate merge of a beam step is one small launch (isc_beam_select, pointwise.hip);
//   * the gated attention scan runs one 1024-thread workgroup per row with every region's loads in flight at once.
// All arithmetic is exact fp32 FMA in a fixed order.
#include <atomic>

#include "common.h"

#define ROWS_MAX 8
#define ROWS_FIN_MAX 4       // isc_rows_ext.fin_prev: rows up to which the previous step's finalize rides on the att-LSTM launch
#define ROWS_MAX_SLICE 12
#define ROWS_KC 8                    // candidate slots per (row, column tile)
#if ROWS_STAMP
RSTAMP_SETTER(isc_rows_set_stamp)
#endif

struct RSlice {                      // one K-slice (<= 256 floats) of a contraction
    const float *A;                  // activations: row m at A + row(m) * lda   (already offset to the slice's k0)
    const float *W;                  // weights: row n at W + n * ldw             (already offset to the slice's k0)
    int lda, ldw, klen, indirect;    // indirect: row(m) = src[m] (recurrent state behind the beam re-ordering)
};
// The contraction as the host passes it: up to three K-segments, cut into slices of 1 << cut_shift floats by the device.
// A kernel reads these fields at constant kernarg offsets (one batch of scalar loads, one wait); a table of slices
// indexed by the wave number cost a second, dependent scalar-load round trip in front of the first weight load
// (2.8 us from wave start to the first weight load in the stamp build, tools/rows_stamp_lab.hip).
struct RSeg {
    const float *A, *W;
    int lda, ldw, K, indirect;
};
struct RSegs {
    RSeg s[3];
    int nseg, nslice, cut_shift, pad;
};
__device__ __forceinline__ RSlice rows_slice_of(const RSegs &g, int si) {
    const int cut = 1 << g.cut_shift;
    const int n0 = (g.s[0].K + cut - 1) >> g.cut_shift;
    const int n1 = g.nseg > 1 ? (g.s[1].K + cut - 1) >> g.cut_shift : 0;
    const int sg = si < n0 ? 0 : (si < n0 + n1 ? 1 : 2);
    const int k0 = (sg == 0 ? si : sg == 1 ? si - n0 : si - n0 - n1) << g.cut_shift;
    RSlice r;
    const float *A = sg == 0 ? g.s[0].A : sg == 1 ? g.s[1].A : g.s[2].A;
    const float *W = sg == 0 ? g.s[0].W : sg == 1 ? g.s[1].W : g.s[2].W;
    const int K = sg == 0 ? g.s[0].K : sg == 1 ? g.s[1].K : g.s[2].K;
    r.A = A + k0; r.W = W + k0;
    r.lda = sg == 0 ? g.s[0].lda : sg == 1 ? g.s[1].lda : g.s[2].lda;
    r.ldw = sg == 0 ? g.s[0].ldw : sg == 1 ? g.s[1].ldw : g.s[2].ldw;
    r.indirect = sg == 0 ? g.s[0].indirect : sg == 1 ? g.s[1].indirect : g.s[2].indirect;
    r.klen = K - k0 < cut ? K - k0 : cut;
    return r;
}
#define ROWS_SK 256                  // floats per (slice, row) slot of the staged activations in LDS: 1 KB = one LDS-DMA

typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ldw4(const float *p) {
    if constexpr (NT) {
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4 *>(p);
}

// The weights of NP weight rows over one slice: NP x NSTEP 16-byte loads per lane (k = 64 t + 4 (lane & 15) + 0..3), issued
// back to back, unconditional: a lane past the slice's end reads the slice's last float4 again and meets a staged zero.
// (passes [P0, P1): a kernel issues its first pass ahead of the barrier that publishes the activations and the rest
// behind it - the memory pipeline accepts loads no faster than it serves them, so "issue everything, then the barrier"
// reaches the barrier when most weights have already arrived and the FMAs run after the stream instead of under it)
template <int NP, int NSTEP, bool NT, int P0 = 0, int P1 = NP>
__device__ __forceinline__ void rows_issue(const RSlice &s, const long long (&wrow)[NP], int lane, float4 (&w)[NP][NSTEP]) {
    const int l = lane & 15;
    int kc[NSTEP];
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
        const int kk = t * 64 + l * 4;
        kc[t] = kk < s.klen ? kk : s.klen - 4;
    }
#pragma unroll
    for (int p = P0; p < P1; ++p)
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) w[p][t] = ldw4<NT>(s.W + wrow[p] * s.ldw + kc[t]);
}

// acc[p][m] += sum_k w[p][k] * As[m][k] against the slice's staged activation rows As [MR][ROWS_SK] (zeros past the slice's
// end and in rows >= M); the four 16-lane groups read the same addresses (LDS broadcast).
// Two partial sums per (p, m) - elements 0, 2 of every float4 in .x, elements 1, 3 in .y - so that the contraction is
// packed FMAs (v_pk_fma_f32: two per instruction); rows_acc_sum adds the pair.
typedef float f2v __attribute__((ext_vector_type(2)));
template <int MR, int NP, int NSTEP>
__device__ __forceinline__ void rows_fma(const float *As, const float4 (&w)[NP][NSTEP], int lane, f2v (&acc)[NP][MR]) {
    const int l = lane & 15;
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
        // (one step's activation reads at a time: hoisted together, the MR x NSTEP float4 spilled next to the weights)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");                       // (IR passes move LDS loads across the intrinsic alone)
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const float4 x = *reinterpret_cast<const float4 *>(As + m * ROWS_SK + t * 64 + l * 4);
            const f2v x01 = {x.x, x.y}, x23 = {x.z, x.w};
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const f2v w01 = {w[p][t].x, w[p][t].y}, w23 = {w[p][t].z, w[p][t].w};
                f2v v = acc[p][m];
                v = __builtin_elementwise_fma(w01, x01, v);
                v = __builtin_elementwise_fma(w23, x23, v);
                acc[p][m] = v;
            }
        }
        // (this step's sums pass through an empty asm statement: otherwise every step's FMAs sink below the last step's
        // reads, behind the wait for the weights, and MR x NSTEP activation float4 are live at once - spills at MR = 8)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(acc[p][m]));
    }
}

// The activation slices in LDS: As[(si * MR + m) * ROWS_SK + k] = (m < M && k < klen) ? A_si[row(m), k] : 0, one LDS-DMA
// (64 lanes x 16 B = one 1 KB slot; no staging registers) per (slice, row); lanes that must deliver zeros read a 16-byte
// zero constant.  row(m) = srcv's lane m for the slices of the recurrent state when the caller re-orders it.
// The J waves that contract slice si stage it between them (wave j: rows j, j + J, ...) as the FIRST vector-memory
// operations they issue: the requests sit at the head of every queue (a helper wave's DMAs, issued beside 96 KB of weight
// loads per CU, took 2.5 - 3.5 us to land: stamps, tools/stamp_step.py), and since vmcnt retires in order the wave can
// wait for exactly these - s_waitcnt vmcnt(<its weight loads>) - while its weights stay in flight.
struct __attribute__((aligned(16))) RowsConst {
    long long ident[ROWS_MAX];
    float zero[4];
};
__device__ RowsConst g_rows_const = {{0, 1, 2, 3, 4, 5, 6, 7}, {0.f, 0.f, 0.f, 0.f}};

// srcv: lane q (< 8) holds the source row of slot q (readlane per slot: the slot index is wave-uniform).
template <int MR>
__device__ __forceinline__ void rows_stage_own(const RSegs &g, int si0, int S, int j, int J, int srcv, int M, float *As,
                                               const RowsConst *rc, int lane) {
    const unsigned lds0 = (unsigned)(size_t)As;
    for (int si = si0; si < g.nslice; si += S) {             // wave-uniform
        const RSlice s = rows_slice_of(g, si);
        for (int m = j; m < MR; m += J) {
            const long long r = s.indirect ? __builtin_amdgcn_readlane(srcv, m) : m;
            const bool valid = m < M && lane * 4 < s.klen;
            const float *src = valid ? s.A + r * s.lda + lane * 4 : rc->zero;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(si * MR + m) * (ROWS_SK * 4));
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(dst), "v"(src) : "memory");      // (M0: only these statements write it in this file)
        }
    }
}
// Wait for the wave's DMAs with its NW younger weight loads still in flight.
template <int NW>
__device__ __forceinline__ void rows_stage_wait() {
    static_assert(NW >= 0 && NW < 64, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NW) : "memory");
}

static const RowsConst *rows_const() {              // device address of the constants (per process: one device code object)
    static std::atomic<const RowsConst *> p{nullptr};
    const RowsConst *v = p.load();
    if (!v) {
        void *q = nullptr;
        if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_rows_const)) != hipSuccess) return nullptr;
        v = static_cast<const RowsConst *>(q);
        p.store(v);
    }
    return v;
}
// Cache policy of the weight streams.  A decode step reads 44.6 MB of fp32 weights; the eight XCD L2s hold 32 MB, and
// they DO keep read-only lines from one launch to the next (tools/cache_sweep_lab.hip: a 24 MB set re-read by the same
// workgroups every launch runs at 14 TB/s; 48 MB cycled through them at 6.6 TB/s, i.e. from the Infinity Cache).  So the
// classifier's 20.5 MB - read once per step, the largest single stream - is loaded non-temporally (it passes without
// evicting anything: tools/cache_policy_lab.hip, the resident set's sweep 1.7 us with or without it, 2.75 us when the
// stream allocates), and the 24.1 MB of the two LSTM cells and the projections stay L2-resident across steps.
// 0 = every stream default policy, 1 = classifier non-temporal (default), 2 = every stream non-temporal.
static std::atomic<int> g_rows_nt{1};
extern "C" int isc_set_rows_nt(int mode) { return g_rows_nt.exchange(mode < 0 ? 0 : mode > 2 ? 2 : mode); }
static std::atomic<long long> g_rows_launches{0};
extern "C" long long isc_rows_launches(void) { return g_rows_launches.load(); }
#define ROWS_LDS_MAX 110000

// ------------------------------------------------------------------ LSTM cell
// gates[m, g*H + u] = sum over slices; c' = sig(f) c + sig(i) tanh(g), h' = sig(o) tanh(c')  (captioner.py:175,181)
// Workgroup = S x J streaming waves + 1 epilogue wave.  Streaming wave (sw, j): K-slices sw, sw + S, ... of units
// u0 + j*UPW .. +UPW-1 (lane group g = gate g of the unit): its share of the activation image by LDS-DMA, weight loads,
// barrier, FMAs against the staged rows, 16-lane sums, partials to LDS.  The epilogue wave holds the dependent index
// chain (token id -> row of the token table) and fetches c_prev / hoisted term / biases of the workgroup's U = J*UPW
// units at kernel start; it applies the cell once the partials are in LDS.
struct RLstmArgs {
    RSegs g;
    int S, J, has_src, M, H, row_div;   // row_div: rows per image of the per-image hoisted term `pre` (1: per row)
    const int *skip;                 // optional: *skip == 0 -> nothing to do (the search has ended; see isc_rows_ext.live_in)
    const long long *src;            // never null (identity when the caller has no re-ordering)
    const RowsConst *rc;
    const float *c_prev;
    float *h_out, *c_out;
    const float *b_ih, *b_hh, *pre, *tab;
    const long long *tab_ids;
    long long tab_ids_stride;
    // optional (pmax != null): the PREVIOUS decode step's roll-out finalize (isc_rollout_finalize, greedy form) done by
    // this launch - see isc_rows_ext.fin_prev
    struct Fin {
        const float *pmax, *psum;
        const int *pidx;
        int n_tile, T, t;
        long long eos;
        long long *seq, *raw;
        float *lp, *masks;
        const int *unf_in;
        int *unf_out, *alive;
    } fin;
};
ISC_STATUS_DECL(rows)

template <int MR, int UPW, bool NT>
__global__ __launch_bounds__(768) void rows_lstm_kernel(const RLstmArgs a) {
    rows_kernarg_warm<ROWS_KERNARG_LINES(RLstmArgs)>();
    extern __shared__ __attribute__((aligned(16))) float As[];                  // [nslice][MR][ROWS_SK]
    __shared__ float red[8 * 2 * 4 * ROWS_MAX];             // [S*J waves][UPW][4 gates][MR]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // an SGPR: slice descriptors by scalar loads
    const int S = a.S, J = a.J, NWS = S * J, H = a.H, M = a.M;
    const int U = J * UPW, u0 = blockIdx.x * U;
    const int run = a.skip ? *a.skip : 1;                   // (uniform; waited for at the first barrier, not before)
#define RSTAMP_KID (a.b_ih ? 3 : 0)
    RSTAMP(0);
    if (wave < NWS) {
        const int sw = wave % S, j = wave / S;
        long long wrow[UPW];
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int unit = u0 + j * UPW + i;
            wrow[i] = (long long)g * H + (unit < H ? unit : H - 1);
        }
        f2v acc2[UPW][MR];
#pragma unroll
        for (int i = 0; i < UPW; ++i)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc2[i][m] = f2v{0.f, 0.f};
        float4 w[UPW][4];
        int si = sw;
        const RSlice s0 = rows_slice_of(a.g, si < a.g.nslice ? si : 0);
        if (run == 0) return;
        RSTAMP(1);
        // source rows of the recurrent state (only when the caller re-orders it: a dependent load in front of the DMAs)
        int srcv = lane & 7;
        if (a.has_src) srcv = (int)a.src[(lane & 7) < M ? (lane & 7) : M - 1];
        rows_stage_own<MR>(a.g, sw, S, j, J, srcv, M, As, a.rc, lane);
        if (si < a.g.nslice) rows_issue<UPW, 4, NT>(s0, wrow, lane, w);
        RSTAMP(2);
        rows_stage_wait<UPW * 4>();
        __syncthreads();                                    // the staged activations
        RSTAMP(3);
        while (si < a.g.nslice) {
            rows_fma<MR, UPW, 4>(As + si * MR * ROWS_SK, w, lane, acc2);
            si += S;
            if (si < a.g.nslice) rows_issue<UPW, 4, NT>(rows_slice_of(a.g, si), wrow, lane, w);
        }
        float acc[UPW][MR];
#pragma unroll
        for (int i = 0; i < UPW; ++i)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[i][m] = row16_sum(acc2[i][m].x + acc2[i][m].y);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int i = 0; i < UPW; ++i)
#pragma unroll
                for (int m = 0; m < MR; ++m) red[((wave * UPW + i) * 4 + g) * MR + m] = acc[i][m];
        }
        RSTAMP(4);
        __syncthreads();
        return;
    }
    // ---- epilogue wave: lane e = (unit ul = e / MR of the workgroup, row m = e % MR)
    const int ul = lane / MR, m = lane - ul * MR, unit = u0 + ul;
    const bool ok = ul < U && m < M && unit < H;
    long long tok = 0;
    if (MR <= ROWS_FIN_MAX && a.fin.pmax) {    // (compiled for up to four rows: with more the fold outlasts the launch it saves)
        // The previous step's finalize (captioner.py:329-344, greedy): its tile statistics folded here, by the wave that
        // waits for the partial sums anyway - every workgroup derives the tokens it feeds (rows x 250 tiles, L2-hot),
        // workgroup 0 writes that step's outputs.  The unfinished flags are read from one array and written to another:
        // the other workgroups read them while workgroup 0 updates.  Same fold, same arithmetic as the finalize kernel.
        const RLstmArgs::Fin &F = a.fin;
        int alive_next = 0, my_tok = 0;
        const int alive_t = F.alive[F.t];
        if (F.n_tile <= 256) {
            // every row's statistics requested at once, the rows' reductions side by side (independent chains), the stores
            // last: the fold is over before the activations of this step are staged - one row after the other, with the
            // writer's stores in between, it took longer than the finalize launch it replaces
            float pmv[MR][4], psv[MR][4];
            int piv[MR][4], un[MR];
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                const long long so = (long long)(r < M ? r : M - 1) * F.n_tile;
                un[r] = F.unf_in[r < M ? r : M - 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = lane + 64 * q, ic = i < F.n_tile ? i : F.n_tile - 1;
                    pmv[r][q] = F.pmax[so + ic]; psv[r][q] = F.psum[so + ic]; piv[r][q] = F.pidx[so + ic];
                }
            }
            float mx[MR], sm[MR];
            int ix[MR];
#pragma unroll
            for (int r = 0; r < MR; ++r) {                   // (fold_row_stats_impl, with the loads hoisted)
                mx[r] = -INFINITY; ix[r] = 0x7fffffff;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (lane + 64 * q < F.n_tile) {
                        const float v = pmv[r][q];
                        const int id = piv[r][q];
                        if (v > mx[r] || (v == mx[r] && id < ix[r])) { mx[r] = v; ix[r] = id; }
                    }
            }
#pragma unroll
            for (int r = 0; r < MR; ++r) half_argmax(mx[r], ix[r]);
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                const float ov = __shfl_xor(mx[r], 32, 64);
                const int oi = __shfl_xor(ix[r], 32, 64);
                if (ov > mx[r] || (ov == mx[r] && oi < ix[r])) { mx[r] = ov; ix[r] = oi; }
            }
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (lane + 64 * q < F.n_tile) s += psv[r][q] * expf(pmv[r][q] - mx[r]);
                sm[r] = half_sum(s);
            }
#pragma unroll
            for (int r = 0; r < MR; ++r) sm[r] += __shfl_xor(sm[r], 32, 64);
            const bool writer = blockIdx.x == 0 && lane == 0 && alive_t != 0;    // (alive == 0: the reference has left its loop)
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                if (r < M) {                                 // (uniform)
                    if (lane == 0 && !(fabsf(mx[r]) <= 3.0e38f && sm[r] <= 3.0e38f)) isc_flag_rows(ISC_STATUS_WORD_STATS);
                    const int gidx = ix[r] == 0x7fffffff ? 0 : ix[r];
                    const long long itm = un[r] ? gidx : 0;  // finished rows feed <PAD> (id 0): `it * unfinished`
                    const int u2 = un[r] && (itm != F.eos);
                    alive_next += u2;
                    my_tok = m == r ? (int)itm : my_tok;
                    if (writer) {
                        const long long o = (long long)r * F.T + F.t;
                        F.masks[o] = (float)un[r];
                        F.seq[o] = itm;
                        F.lp[o] = -logf(sm[r]);              // log_softmax at the arg-max: (x_max - x_max) - log S
                        if (F.raw) F.raw[o] = gidx;
                        F.unf_out[r] = u2;
                    }
                }
            }
            if (writer && alive_next) F.alive[F.t + 1] = alive_next;
        } else {
            const bool writer = blockIdx.x == 0 && lane == 0 && alive_t != 0;
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                if (r < M) {                                 // (uniform)
                    float gmax, S;
                    int gidx;
                    const long long so = (long long)r * F.n_tile;
                    if (fold_row_stats_impl(F.pmax + so, F.psum + so, F.pidx + so, F.n_tile, lane, gmax, gidx, S) && lane == 0)
                        isc_flag_rows(ISC_STATUS_WORD_STATS);
                    const int u = F.unf_in[r];
                    const long long itm = u ? gidx : 0;
                    const int u2 = u && (itm != F.eos);
                    alive_next += u2;
                    my_tok = m == r ? (int)itm : my_tok;
                    if (writer) {
                        const long long o = (long long)r * F.T + F.t;
                        F.masks[o] = (float)u;
                        F.seq[o] = itm;
                        F.lp[o] = -logf(S);
                        if (F.raw) F.raw[o] = gidx;
                        F.unf_out[r] = u2;
                    }
                }
            }
            if (writer && alive_next) F.alive[F.t + 1] = alive_next;
        }
        tok = my_tok;
    } else if (ok && a.tab) {
        tok = a.tab_ids[(long long)m * a.tab_ids_stride];
    }
    long long rs = m;
    if (ok && a.has_src) rs = a.src[m];
    if (run == 0) return;
    RSTAMP(1);
    float cp = 0.f, q[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        cp = a.c_prev[rs * H + unit];
        if (a.pre) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = a.pre[(long long)(m / a.row_div) * 4 * H + k * H + unit];
        }
        if (a.b_ih) {
#pragma unroll
            for (int k = 0; k < 4; ++k) b[k] = a.b_ih[k * H + unit] + a.b_hh[k * H + unit];
        }
        if (a.tab) {
#pragma unroll
            for (int k = 0; k < 4; ++k) tb[k] = a.tab[tok * 4 * H + k * H + unit];
        }
    }
    RSTAMP(3);
    __syncthreads();                                        // activations published
    __syncthreads();                                        // partial sums in LDS
    RSTAMP(5);
    if (!ok) return;
    const int j = ul / UPW, i = ul - j * UPW;
    float gt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float pr[8];                                        // the slices' partials, read together (S <= 8)
#pragma unroll
        for (int sw = 0; sw < 8; ++sw) pr[sw] = red[(((j * S + (sw < S ? sw : 0)) * UPW + i) * 4 + k) * MR + m];
        float v = pr[0];
#pragma unroll
        for (int sw = 1; sw < 8; ++sw) v += sw < S ? pr[sw] : 0.f;
        // g += (b_ih + b_hh);  g += pre;  g += table row   (the order of lstm_cells, gemm_f32.hip)
        if (a.b_ih) v += b[k];
        if (a.pre) v += q[k];
        if (a.tab) v += tb[k];
        gt[k] = v;
    }
    const float gi = isc_sigmoid(gt[0]), gf = isc_sigmoid(gt[1]), gg = isc_tanh(gt[2]), go = isc_sigmoid(gt[3]);
    const float c2 = gf * cp + gi * gg;
    const float h2 = go * isc_tanh(c2);
    a.c_out[(long long)m * H + unit] = c2;
    a.h_out[(long long)m * H + unit] = h2;
    RSTAMP(6);
#undef RSTAMP_KID
}

// The segment list of one [M, K] x [N, K]^T contraction, cut at 1 << cut_shift floats.  Returns the slice count or -1.
static int rows_make_segs(RSegs &g, const float *const *A, const int *lda, const float *const *W, const int *ldw,
                          const int *K, const int *indirect, int nseg, int cut_shift) {
    if (nseg < 1 || nseg > 3) return -1;
    g = RSegs{};
    int n = 0;
    for (int s = 0; s < nseg; ++s) {
        if (K[s] <= 0 || (K[s] & 3) || (lda[s] & 3) || (ldw[s] & 3) || !A[s] || !W[s]) return -1;
        if (!isc_aligned16(A[s]) || !isc_aligned16(W[s])) return -1;
        g.s[s].A = A[s]; g.s[s].W = W[s]; g.s[s].lda = lda[s]; g.s[s].ldw = ldw[s]; g.s[s].K = K[s];
        g.s[s].indirect = indirect ? indirect[s] : 0;
        n += (K[s] + (1 << cut_shift) - 1) >> cut_shift;
    }
    for (int s = nseg; s < 3; ++s) g.s[s] = g.s[0];      // never selected; valid pointers all the same
    if (n > ROWS_MAX_SLICE) return -1;
    g.nseg = nseg; g.nslice = n; g.cut_shift = cut_shift;
    return n;
}

static int rows_mr(int M) { return M <= 2 ? 2 : M <= 4 ? 4 : M <= 6 ? 6 : 8; }

template <typename K>
static int rows_lds_attr(K kernel, std::atomic<bool> &done) {
    if (!done.load()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           ROWS_LDS_MAX);
        if (e != hipSuccess) return (int)e;
        done.store(true);
    }
    return ISC_OK;
}
#define ROWS_LAUNCH(KERNEL, GRID, THREADS, LDS, ST, ARGS)                        \
    do {                                                                        \
        static std::atomic<bool> attr_done{false};                              \
        int rc_ = rows_lds_attr(&KERNEL, attr_done);                            \
        if (rc_) return rc_;                                                    \
        hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(THREADS), LDS, ST, ARGS);   \
    } while (0)

template <int UPW, bool NT>
static int rows_lstm_launch(const RLstmArgs &a, int grid, int threads, size_t lds, hipStream_t st) {
    switch (rows_mr(a.M)) {
        case 2: ROWS_LAUNCH((rows_lstm_kernel<2, UPW, NT>), grid, threads, lds, st, a); break;
        case 4: ROWS_LAUNCH((rows_lstm_kernel<4, UPW, NT>), grid, threads, lds, st, a); break;
        case 6: ROWS_LAUNCH((rows_lstm_kernel<6, UPW, NT>), grid, threads, lds, st, a); break;
        default: ROWS_LAUNCH((rows_lstm_kernel<8, UPW, NT>), grid, threads, lds, st, a); break;
    }
    return ISC_OK;
}

// segs: (A, lda, W, ldw, K, indirect) x nseg.
static int rows_lstm(const float *const *A, const int *lda, const float *const *W, const int *ldw, const int *K,
                     const int *ind, int nseg, int M, int H, const int64_t *src, const float *c_prev, float *h_out,
                     float *c_out, const float *b_ih, const float *b_hh, const float *pre, const float *tab,
                     const int64_t *tab_ids, int64_t tab_ids_stride, int row_div, const int *skip, hipStream_t st,
                     const RLstmArgs::Fin *fin = nullptr) {
    RLstmArgs a = {};
    if (fin) a.fin = *fin;
    const int nslice = rows_make_segs(a.g, A, lda, W, ldw, K, ind, nseg, 8);
    if (nslice < 1) return ISC_E_SHAPE;
    if (M < 1 || M > ROWS_MAX || H < 1) return ISC_E_SHAPE;
    if (!c_prev || !h_out || !c_out) return ISC_E_NULL;
    if ((b_ih == nullptr) != (b_hh == nullptr)) return ISC_E_NULL;
    if (tab && !tab_ids) return ISC_E_NULL;
    const int MR = rows_mr(M);
    a.S = nslice < 8 ? nslice : 8;
    // units per workgroup U = J * UPW: about 256 workgroups, at most 8 streaming waves, U * MR <= 64 epilogue lanes
    int want = (H + 255) / 256;                       // units per workgroup that fill 256 CUs once
    if (want < 1) want = 1;
    int J = 8 / a.S;
    if (J < 1) J = 1;
    int upw = 1;
    if (J > want) J = want;
    if (J < want) upw = 2;
    while (J > 1 && J * upw * MR > 64) --J;
    if (J * upw * MR > 64) upw = 1;
    a.J = J;
    a.has_src = src != nullptr;
    a.M = M; a.H = H; a.row_div = row_div > 1 ? row_div : 1; a.skip = skip;
    a.rc = rows_const();
    if (!a.rc) return ISC_E_STATE;
    a.src = src ? reinterpret_cast<const long long *>(src) : a.rc->ident;
    a.c_prev = c_prev; a.h_out = h_out; a.c_out = c_out; a.b_ih = b_ih; a.b_hh = b_hh; a.pre = pre; a.tab = tab;
    a.tab_ids = reinterpret_cast<const long long *>(tab_ids); a.tab_ids_stride = tab_ids_stride;
    const int U = J * upw, grid = (H + U - 1) / U, threads = 64 * (a.S * J + 1);
    const size_t lds = (size_t)nslice * MR * ROWS_SK * sizeof(float);
    if (lds > ROWS_LDS_MAX) return ISC_E_SHAPE;
    int rc;
    if (g_rows_nt.load() >= 2) rc = upw == 2 ? rows_lstm_launch<2, true>(a, grid, threads, lds, st) : rows_lstm_launch<1, true>(a, grid, threads, lds, st);
    else rc = upw == 2 ? rows_lstm_launch<2, false>(a, grid, threads, lds, st) : rows_lstm_launch<1, false>(a, grid, threads, lds, st);
    if (rc) return rc;
    ISC_LAUNCH_CHECK();
    ++g_rows_launches;
    return ISC_OK;
}

// ------------------------------------------------------------------ grouped projections of h (h2att, h2word, gate)
// C_i[m, n] = sum_k h[m, k] W_i[n, k] + b_i[n] for up to 3 problems that share the activation rows; a workgroup
// takes 4 * J consecutive output columns of the concatenated [N_0 + N_1 + N_2] axis (N_i % 4 == 0); S x J waves.
struct RLinArgs {
    RSegs g;                         // .W is problem 0's; problem i's = W[i] + (slice.W - W[0])
    int S, J, M, nprob;
    const int *skip;
    const RowsConst *rc;
    const float *W[3], *bias[3];
    float *C[3];
    int N[3], ldc[3];
};

template <int MR>
__global__ __launch_bounds__(768) void rows_linear_kernel(const RLinArgs a) {
    rows_kernarg_warm<ROWS_KERNARG_LINES(RLinArgs)>();
    extern __shared__ __attribute__((aligned(16))) float As[];
    __shared__ float red[8 * 4 * ROWS_MAX];                 // [S*J waves][4 rows][MR]
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.S, J = a.J, M = a.M;
    if (a.skip && *a.skip == 0) return;                     // (uniform)
    const int sw = wave % S, j = wave / S;
    const int n0 = (blockIdx.x * J + j) * 4;                // this wave's 4 output columns (concatenated axis)
    int pi = 0, nl = n0;
    if (a.nprob > 1 && nl >= a.N[0]) { nl -= a.N[0]; pi = 1; }
    if (a.nprob > 2 && pi == 1 && nl >= a.N[1]) { nl -= a.N[1]; pi = 2; }
    const int Np = pi == 0 ? a.N[0] : pi == 1 ? a.N[1] : a.N[2];
    const bool live = nl < Np;
    // epilogue operand of lanes (r, m) of the slice-0 waves, fetched ahead of the weight stream
    const float *bp = pi == 0 ? a.bias[0] : pi == 1 ? a.bias[1] : a.bias[2];
    float bias_v = 0.f;
    if (sw == 0 && live && lane < 4 * MR && bp) bias_v = bp[nl + lane / MR];
    long long wrow[1] = {(long long)((live ? nl : 0) + g)};
    f2v acc2[1][MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc2[0][m] = f2v{0.f, 0.f};
    float4 w[1][4];
    int si = sw;
    const float *Wp = pi == 0 ? a.W[0] : pi == 1 ? a.W[1] : a.W[2];
    RSlice s = rows_slice_of(a.g, si < a.g.nslice ? si : 0);
    s.W = Wp + (s.W - a.W[0]);                              // the slice's k-offset inside problem pi's weights
    rows_stage_own<MR>(a.g, sw, S, j, J, lane & 7, M, As, a.rc, lane);
    if (si < a.g.nslice) rows_issue<1, 4, false>(s, wrow, lane, w);
    rows_stage_wait<4>();
    __syncthreads();
    while (si < a.g.nslice) {
        rows_fma<MR, 1, 4>(As + si * MR * ROWS_SK, w, lane, acc2);
        si += S;
        if (si < a.g.nslice) {
            s = rows_slice_of(a.g, si);
            s.W = Wp + (s.W - a.W[0]);
            rows_issue<1, 4, false>(s, wrow, lane, w);
        }
    }
    float acc[1][MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[0][m] = row16_sum(acc2[0][m].x + acc2[0][m].y);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int m = 0; m < MR; ++m) red[(wave * 4 + g) * MR + m] = acc[0][m];
    }
    __syncthreads();
    if (sw == 0 && live && lane < 4 * MR) {
        const int r = lane / MR, m = lane - r * MR;
        if (m < M) {
            float v = red[((j * S) * 4 + r) * MR + m];
            for (int x = 1; x < S; ++x) v += red[((j * S + x) * 4 + r) * MR + m];
            v += bias_v;
            float *Cp = pi == 0 ? a.C[0] : pi == 1 ? a.C[1] : a.C[2];
            const int ldc = pi == 0 ? a.ldc[0] : pi == 1 ? a.ldc[1] : a.ldc[2];
            Cp[(long long)m * ldc + nl + r] = v;
        }
    }
}

static int rows_linear3(const float *A, int lda, int K, int M, const float *const *W, const float *const *bias,
                        float *const *C, const int *N, const int *ldc, int nprob, const int *skip, hipStream_t st) {
    if (nprob < 1 || nprob > 3 || M < 1 || M > ROWS_MAX) return ISC_E_SHAPE;
    RLinArgs a = {};
    const int ldw = K, ind = 0;
    // slices of 128: twice the waves per output column, the launch is latency-bound (3 MB of weights)
    const int nslice = rows_make_segs(a.g, &A, &lda, &W[0], &ldw, &K, &ind, 1, K >= 256 ? 7 : 8);
    if (nslice < 1) return ISC_E_SHAPE;
    int Ntot = 0;
    for (int i = 0; i < nprob; ++i) {
        if (!W[i] || !C[i]) return ISC_E_NULL;
        if (N[i] <= 0 || (N[i] & 3) || !isc_aligned16(W[i])) return ISC_E_SHAPE;
        a.W[i] = W[i]; a.bias[i] = bias[i]; a.C[i] = C[i]; a.N[i] = N[i]; a.ldc[i] = ldc[i];
        Ntot += N[i];
    }
    a.nprob = nprob; a.M = M; a.skip = skip;
    a.rc = rows_const();
    if (!a.rc) return ISC_E_STATE;
    a.S = nslice < 8 ? nslice : 8;
    int J = 8 / a.S;
    if (J < 1) J = 1;
    while (J > 1 && (Ntot / 4 + J - 1) / J < 192) --J;     // enough workgroups for the chip before fat ones
    a.J = J;
    const int grid = (Ntot / 4 + J - 1) / J, threads = 64 * a.S * J, MR = rows_mr(M);
    const size_t lds = (size_t)nslice * MR * ROWS_SK * sizeof(float);
    if (lds > ROWS_LDS_MAX) return ISC_E_SHAPE;
    switch (MR) {
        case 2: ROWS_LAUNCH((rows_linear_kernel<2>), grid, threads, lds, st, a); break;
        case 4: ROWS_LAUNCH((rows_linear_kernel<4>), grid, threads, lds, st, a); break;
        case 6: ROWS_LAUNCH((rows_linear_kernel<6>), grid, threads, lds, st, a); break;
        default: ROWS_LAUNCH((rows_linear_kernel<8>), grid, threads, lds, st, a); break;
    }
    ISC_LAUNCH_CHECK();
    ++g_rows_launches;
    return ISC_OK;
}

// ------------------------------------------------------------------ classifier: tile statistics + tile top-k
// logits[m, c] = h[m, :] . W[c, :] + b[c] over one tile of TW columns per workgroup (TW = isc_rows_stats_tile(V):
// about V / 256, so that one round of workgroups covers the chip).  Per (row, tile): max, arg-max, sum exp(x - max)
// (the statistics isc_vocab_fwd leaves per 128 columns) and, when beam > 0, the ROWS_KC largest MASKED logits
// (<PAD>, <SOS>, <UNK>, the row's last word at -inf: captioner.py:394-399) with their word ids, descending, ties to
// the smaller id - from which isc_beam_select forms the row's top-`beam` without reading any logits.
struct RVocabArgs {
    RSegs g;
    int S, J, M, V, TW, n_tile;
    const int *skip;
    const RowsConst *rc;
    const float *bias;
    float *pmax, *psum;
    int *pidx;
    float *logits;
    long long ld_logits;
    int beam, mask_special, cons;
    long long pad_id, sos_id, unk_id;
    const long long *last_word;
    float *cand_val;
    int *cand_idx;
};

template <int MR, int NP, bool NT>
__global__ __launch_bounds__(768) void rows_vocab_kernel(const RVocabArgs a) {
    constexpr int TWC = 16 * NP;                            // columns a workgroup can hold: 4 (J) x NP passes x 4 lane groups
    rows_kernarg_warm<ROWS_KERNARG_LINES(RVocabArgs)>();
    extern __shared__ __attribute__((aligned(16))) float As[];
    __shared__ float red[8 * NP * 4 * ROWS_MAX];            // [S*J waves][NP][4][MR]
    __shared__ __attribute__((aligned(16))) unsigned long long kk[ROWS_MAX][64];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.S, J = a.J, M = a.M, TW = a.TW, V = a.V;
    const int tile = blockIdx.x, col0 = tile * TW;
    const int tw = V - col0 < TW ? V - col0 : TW;           // valid columns of this tile
#define RSTAMP_KID 4
    RSTAMP(0);
    RSTAMP_CLK0();
    // epilogue operands, fetched ahead of the weight stream: this lane's column bias, its row's last word
    const float bias_c = (lane < tw) ? a.bias[col0 + lane] : 0.f;
    const int lastv = (a.cons && a.last_word) ? (int)a.last_word[(lane & 7) < M ? (lane & 7) : M - 1] : -1;   // lane q: row q's
    if (a.skip && *a.skip == 0) return;                     // (uniform)
    {
        const int sw = wave % S, j = wave / S;
        // pass p of wave j: columns col0 + (p*J + j)*4 + g
        long long wrow[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int c = (p * J + j) * 4 + g;
            wrow[p] = col0 + (c < tw ? c : tw - 1);
        }
        f2v acc2[NP][MR];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc2[p][m] = f2v{0.f, 0.f};
        float4 w[NP][4];
        int si = sw;
        const RSlice s0 = rows_slice_of(a.g, si < a.g.nslice ? si : 0);
        rows_stage_own<MR>(a.g, sw, S, j, J, lane & 7, M, As, a.rc, lane);
        if (si < a.g.nslice) rows_issue<NP, 4, NT>(s0, wrow, lane, w);
        RSTAMP(1);
        rows_stage_wait<NP * 4>();
        __syncthreads();
        RSTAMP(2);
        while (si < a.g.nslice) {
            rows_fma<MR, NP, 4>(As + si * MR * ROWS_SK, w, lane, acc2);
            si += S;
            if (si < a.g.nslice) rows_issue<NP, 4, NT>(rows_slice_of(a.g, si), wrow, lane, w);
        }
        float acc[NP][MR];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[p][m] = row16_sum(acc2[p][m].x + acc2[p][m].y);
        if ((lane & 15) == 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int m = 0; m < MR; ++m) red[((wave * NP + p) * 4 + g) * MR + m] = acc[p][m];
        }
        RSTAMP(3);
        __syncthreads();
        RSTAMP(4);
    }
    // ---- per row: wave m (lane = column c) forms its row of the tile, takes the statistics by lane exchanges, and -
    // beam search - ranks every column of the row: the (masked value, column) pairs become unique unsigned keys
    // (order-preserving bits of the float, then 63 - c: descending value, ties to the lower column), exchanged through
    // LDS by the wave itself (no barrier), one 64-bit compare + one add-with-carry per pair.
    for (int m = wave; m < M; m += nw) {
        const int c = lane;
        float x = -INFINITY, mk = -INFINITY;
        if (c < tw) {
            const int q4 = c >> 2, gg = c & 3, jj = q4 % J, p = q4 / J;
            float v = red[(((jj * S) * NP + p) * 4 + gg) * MR + m];
            for (int s2 = 1; s2 < S; ++s2) v += red[(((jj * S + s2) * NP + p) * 4 + gg) * MR + m];
            const long long id = col0 + c;
            x = v + bias_c;
            if (a.logits) a.logits[(long long)m * a.ld_logits + id] = x;
            bool banned = a.mask_special && (id == a.pad_id || id == a.sos_id || id == a.unk_id);
            if (id == (long long)__builtin_amdgcn_readlane(lastv, m)) banned = true;
            mk = banned ? -INFINITY : x;
        }
        if (a.beam > 0) {
            const unsigned u = __float_as_uint(mk);
            const unsigned key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // ascending with the float order
            kk[m][c] = ((unsigned long long)key << 6) | (unsigned)(63 - c);
        }
        float mx = x;
        int ix = c;
        half_argmax(mx, ix);                                // ties: the lower column
        {
            const float ov = __shfl_xor(mx, 32, 64);
            const int oi = __shfl_xor(ix, 32, 64);
            if (ov > mx || (ov == mx && oi < ix)) { mx = ov; ix = oi; }
        }
        float sm = half_sum(c < tw ? __expf(x - mx) : 0.f);
        sm += __shfl_xor(sm, 32, 64);
        if (lane == 0) {
            const long long o = (long long)m * a.n_tile + tile;
            a.pmax[o] = mx;
            a.psum[o] = sm;
            a.pidx[o] = col0 + ix;
        }
        if (a.beam > 0) {
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): own wave's LDS stores before its loads
            const unsigned long long mine = kk[m][c];
            int rank = 0;
#pragma unroll 1
            for (int ch = 0; ch < NP; ++ch) {                        // 16 keys at a time (32 VGPRs), not all 16 * NP
                unsigned long long kv[16];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const ulonglong2 t2 = *reinterpret_cast<const ulonglong2 *>(&kk[m][16 * ch + 2 * q]);
                    kv[2 * q] = t2.x; kv[2 * q + 1] = t2.y;
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) rank += kv[q] > mine;
            }
            if (c < TWC && rank < ROWS_KC) {
                const long long o = ((long long)m * a.n_tile + tile) * ROWS_KC + rank;
                a.cand_val[o] = mk;
                a.cand_idx[o] = c < tw ? col0 + c : 0;
            }
        }
    }
    RSTAMP(6);
    RSTAMP_CLK1();
#undef RSTAMP_KID
}

extern "C" int isc_rows_stats_tile(int V) {
    if (V <= 0) return 0;
    int tw = 4 * ((V / 4 + 255) / 256);                    // one round of <= 256 workgroups
    if (tw < 16) tw = 16;
    if (tw > 64) tw = 64;
    return tw;
}

static int rows_vocab(const float *h, int ldh, const float *W, int ldw, const float *bias, int M, int V, int K,
                      int TW, float *pmax, float *psum, int *pidx, float *logits, int64_t ld_logits,
                      const isc_rows_ext *x, hipStream_t st) {
    if (!h || !W || !bias || !pmax || !psum || !pidx) return ISC_E_NULL;
    if (M < 1 || M > ROWS_MAX || V < 1 || TW < 16 || TW > 64 || (TW & 3)) return ISC_E_SHAPE;
    RVocabArgs a = {};
    const int ind = 0;
    const int nslice = rows_make_segs(a.g, &h, &ldh, &W, &ldw, &K, &ind, 1, 8);
    if (nslice < 1) return ISC_E_SHAPE;
    a.S = nslice < 8 ? nslice : 8;
    int J = 8 / a.S;
    if (J < 1) J = 1;
    if (J > 4) J = 4;                                      // the kernel ranks a tile of at most 16 * NP columns
    const int passes = (TW + 3) / 4;
    if (J > passes) J = passes;
    int np = (passes + J - 1) / J;
    if (16 * np < TW) np = (TW + 15) / 16;
    if (np > 4) return ISC_E_SHAPE;
    a.J = J; a.M = M; a.V = V; a.TW = TW; a.n_tile = (V + TW - 1) / TW; a.skip = x ? x->live_in : nullptr;
    a.rc = rows_const();
    if (!a.rc) return ISC_E_STATE;
    a.bias = bias; a.pmax = pmax; a.psum = psum; a.pidx = pidx; a.logits = logits; a.ld_logits = ld_logits;
    if (x && x->beam > 0) {
        if (!x->cand_val || !x->cand_idx) return ISC_E_NULL;
        if (x->beam > ROWS_KC) return ISC_E_SHAPE;
        a.beam = x->beam; a.mask_special = x->mask_special; a.cons = x->decoding_constraint;
        a.pad_id = x->pad_id; a.sos_id = x->sos_id; a.unk_id = x->unk_id;
        a.last_word = reinterpret_cast<const long long *>(x->last_word);
        a.cand_val = x->cand_val; a.cand_idx = x->cand_idx;
        if (a.cons && !a.last_word) return ISC_E_NULL;
    }
    const int threads = 64 * a.S * J, MR = rows_mr(M), nt = g_rows_nt.load();
    const size_t lds = (size_t)nslice * MR * ROWS_SK * sizeof(float);
    if (lds > ROWS_LDS_MAX) return ISC_E_SHAPE;
#define RV_LAUNCH(MRV, NPV)                                                                                     \
    do {                                                                                                        \
        if (nt) ROWS_LAUNCH((rows_vocab_kernel<MRV, NPV, true>), a.n_tile, threads, lds, st, a);                  \
        else ROWS_LAUNCH((rows_vocab_kernel<MRV, NPV, false>), a.n_tile, threads, lds, st, a);                    \
    } while (0)
#define RV_NP(MRV)                                                                          \
    do {                                                                                    \
        if (np <= 1) RV_LAUNCH(MRV, 1); else if (np == 2) RV_LAUNCH(MRV, 2);                \
        else if (np == 3) RV_LAUNCH(MRV, 3); else RV_LAUNCH(MRV, 4);                        \
    } while (0)
    if (MR == 2) RV_NP(2); else if (MR == 4) RV_NP(4); else if (MR == 6) RV_NP(6); else RV_NP(8);
#undef RV_NP
#undef RV_LAUNCH
    ISC_LAUNCH_CHECK();
    ++g_rows_launches;
    return ISC_OK;
}

// ------------------------------------------------------------------ gated attention scan, 1024 threads per row
// captioner.py:96-118 as in attn_scan_gate_kernel (attention.hip): content scan over the R regions, sentiment scan over
// the Mw words, z = zh + cont2att(v) + senti2att(s) from the pre-projected rows G, beta, f = beta v + (1 - beta) s.
// Sixteen waves: waves 0-11 take content regions r = wave, wave + 12, ...; waves 12-15 the sentiment words.  With
// R <= 36 and Mw <= 12 (three rows per wave) the P, V and G rows of every region are loaded up front - the row's whole
// 221 + 68 KB in flight at once - and the kernel is one memory round trip plus four workgroup barriers.
struct RScanArgs {
    const float *P[2], *V[2], *G[2];           // [rows, R, A] per row (content), table or per row (sentiment)
    const float *q[2], *q2, *w[2], *wb[2];
    const long long *ids;                      // sentiment gather mode: word row ids [rows, ids_ld]
    long long ids_ld;
    int R[2], A, row_div;                      // row_div: rows per image of the per-image P / V / G / ids / q2 (1: per row)
    const int *skip;
    const float *zh, *b_c, *b_s, *w_g, *b_g;
    float *f, *alpha[2], *beta;
    long long alpha_ld[2], beta_ld;
    _Float16 *f_hi, *f_lo;                     // optional split-f16 planes of f (the MFMA lang-LSTM of larger batches)
};
#define RS_NWC 12
#define RS_NWS 4
#define RS_RPW 3

__global__ __launch_bounds__(1024) void rows_scan_gate_kernel(const RScanArgs a) {
    rows_kernarg_warm<ROWS_KERNARG_LINES(RScanArgs)>();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave >= RS_NWC ? 1 : 0;
    const int wv = half ? wave - RS_NWC : wave, nwv = half ? RS_NWS : RS_NWC;
    const int A = a.A, na4 = A >> 2, R = (half ? a.R[1] : a.R[0]);
    const int Rp0 = (a.R[0] + 3) & ~3, Rp1 = (a.R[1] + 3) & ~3;
    float *sc = smem + (half ? Rp0 : 0);                       // scores -> alphas of this half
    float *part = smem + Rp0 + Rp1;                            // [16 waves][2 (V, G)][A]
    float *fin = part + 16 * 2 * A;                            // [4][A]: v, Gc-sum, s, Gs-sum
    float *redw = fin + 4 * A;                                 // [16] wave partials of the gate dot
    if (a.skip && *a.skip == 0) return;                        // (uniform)
#define RSTAMP_KID 2
    RSTAMP(0);
    const bool gather = half && a.ids;
    const int bi = b / a.row_div;                              // the image of this row: P / V / G / ids / q2 are per image
    const float *Pb = (half ? a.P[1] : a.P[0]), *Vb = (half ? a.V[1] : a.V[0]), *Gb = (half ? a.G[1] : a.G[0]);
    if (!gather) {
        Pb += (long long)bi * R * A; Vb += (long long)bi * R * A; Gb += (long long)bi * R * A;
    }
    const bool single = a.R[0] <= RS_NWC * RS_RPW && a.R[1] <= RS_NWS * RS_RPW;
    float4 pp[RS_RPW][2], vv[RS_RPW][2], gg[RS_RPW][2];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // Load order = vmcnt order: word ids (sentiment waves), then q / q2 / w (small, cache-resident), then the rows - a use of
    // q waits for nothing younger than itself.  Every load is unconditional (lanes past A re-read the last float4 and carry
    // w = 0; their sums are never stored): no exec-masked region splits the burst.
    int a4c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a4c[i] = lane + 64 * i < na4 ? lane + 64 * i : na4 - 1;
    long long rrow[RS_RPW];
#pragma unroll
    for (int u = 0; u < RS_RPW; ++u) {
        const int r = wv + u * nwv;
        const int rc = r < R ? r : R - 1;
        rrow[u] = gather ? a.ids[(long long)bi * a.ids_ld + rc] : rc;
    }
    float4 qv[2], q2v[2], wv4[2];
    const bool has_q2 = half && a.q2;                          // wave-uniform
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        qv[i] = reinterpret_cast<const float4 *>((half ? a.q[1] : a.q[0]) + (long long)b * A)[a4c[i]];
        q2v[i] = has_q2 ? reinterpret_cast<const float4 *>(a.q2 + (long long)bi * A)[a4c[i]] : z4;
        wv4[i] = reinterpret_cast<const float4 *>((half ? a.w[1] : a.w[0]))[a4c[i]];
    }
    const float *wbp = half ? a.wb[1] : a.wb[0];
    const float w_bias = wbp ? wbp[0] : 0.f;
    // first (only, when `single`) round of row loads
#pragma unroll
    for (int u = 0; u < RS_RPW; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) pp[u][i] = reinterpret_cast<const float4 *>(Pb + rrow[u] * A)[a4c[i]];
    if (single) {
#pragma unroll
        for (int u = 0; u < RS_RPW; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                vv[u][i] = reinterpret_cast<const float4 *>(Vb + rrow[u] * A)[a4c[i]];
                gg[u][i] = reinterpret_cast<const float4 *>(Gb + rrow[u] * A)[a4c[i]];
            }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        qv[i].x += q2v[i].x; qv[i].y += q2v[i].y; qv[i].z += q2v[i].z; qv[i].w += q2v[i].w;
        if (lane + 64 * i >= na4) wv4[i] = z4;
    }
    RSTAMP(1);
    // ---- scores
    for (int r0 = wv; r0 < R; r0 += nwv * RS_RPW) {
        if (r0 != wv) {
#pragma unroll
            for (int u = 0; u < RS_RPW; ++u) {
                const int r = r0 + u * nwv;
                const int rc = r < R ? r : R - 1;
                const long long row = gather ? a.ids[(long long)bi * a.ids_ld + rc] : rc;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    pp[u][i] = reinterpret_cast<const float4 *>(Pb + row * A)[a4c[i]];
                }
            }
        }
        float acc[RS_RPW];
#pragma unroll
        for (int u = 0; u < RS_RPW; ++u) {
            acc[u] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {                  // lanes past A carry w = 0
                acc[u] += wv4[i].x * isc_tanh(pp[u][i].x + qv[i].x);
                acc[u] += wv4[i].y * isc_tanh(pp[u][i].y + qv[i].y);
                acc[u] += wv4[i].z * isc_tanh(pp[u][i].z + qv[i].z);
                acc[u] += wv4[i].w * isc_tanh(pp[u][i].w + qv[i].w);
            }
        }
#pragma unroll
        for (int u = 0; u < RS_RPW; ++u) acc[u] = half_sum(acc[u]);
#pragma unroll
        for (int u = 0; u < RS_RPW; ++u) acc[u] += __shfl_xor(acc[u], 32, 64);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < RS_RPW; ++u)
                if (r0 + u * nwv < R) sc[r0 + u * nwv] = acc[u] + w_bias;
        }
    }
    // gate operands of threads < A/4: in flight across the softmax and the weighted sums
    float4 zh4 = z4, bc4 = z4, bs4 = z4, wg4 = z4;
    if (tid < na4) {
        zh4 = reinterpret_cast<const float4 *>(a.zh + (long long)b * A)[tid];
        bc4 = reinterpret_cast<const float4 *>(a.b_c)[tid];
        bs4 = reinterpret_cast<const float4 *>(a.b_s)[tid];
        wg4 = reinterpret_cast<const float4 *>(a.w_g)[tid];
    }
    const float b_gate = a.b_g ? a.b_g[0] : 0.f;
    RSTAMP(2);
    __syncthreads();
    RSTAMP(3);
    // ---- softmax statistics of this half's scores: lane r holds score r (64 at a time), wave max / sum by lane exchanges
    // (a per-thread loop over the R scores in LDS was 2 R dependent LDS reads: 3 us of an 11 us kernel at R = 36)
    float mx = -INFINITY;
    for (int r0 = 0; r0 < R; r0 += 64) {
        float v = r0 + lane < R ? sc[r0 + lane] : -INFINITY;
        v = fmaxf(v, isc_dpp<ISC_DPP_XOR1>(v)); v = fmaxf(v, isc_dpp<ISC_DPP_XOR2>(v));
        v = fmaxf(v, isc_dpp<ISC_DPP_HALF_MIRROR>(v)); v = fmaxf(v, isc_dpp<ISC_DPP_MIRROR>(v));
        v = fmaxf(v, isc_swz16(v));
        v = fmaxf(v, __shfl_xor(v, 32, 64));
        mx = fmaxf(mx, v);
    }
    float zs = 0.f;
    for (int r0 = 0; r0 < R; r0 += 64) {
        float e = r0 + lane < R ? __expf(sc[r0 + lane] - mx) : 0.f;
        e = half_sum(e);
        zs += e + __shfl_xor(e, 32, 64);
    }
    const float inv = 1.0f / zs;
    // ---- weighted sums of this wave's rows (ascending region order within the wave)
    float4 o[2] = {z4, z4}, og[2] = {z4, z4};
    for (int r0 = wv; r0 < R; r0 += nwv * RS_RPW) {
        float al[RS_RPW];
#pragma unroll
        for (int u = 0; u < RS_RPW; ++u) {
            const int r = r0 + u * nwv;
            al[u] = r < R ? __expf(sc[r] - mx) * inv : 0.f;
            if (r < R && lane == 0 && (half ? a.alpha[1] : a.alpha[0])) (half ? a.alpha[1] : a.alpha[0])[(long long)b * (half ? a.alpha_ld[1] : a.alpha_ld[0]) + r] = al[u];
        }
        if (!single) {
#pragma unroll
            for (int u = 0; u < RS_RPW; ++u) {
                const int r = r0 + u * nwv;
                const int rc = r < R ? r : R - 1;
                const long long row = gather ? a.ids[(long long)bi * a.ids_ld + rc] : rc;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    vv[u][i] = reinterpret_cast<const float4 *>(Vb + row * A)[a4c[i]];
                    gg[u][i] = reinterpret_cast<const float4 *>(Gb + row * A)[a4c[i]];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RS_RPW; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                o[i].x += al[u] * vv[u][i].x; o[i].y += al[u] * vv[u][i].y;
                o[i].z += al[u] * vv[u][i].z; o[i].w += al[u] * vv[u][i].w;
                og[i].x += al[u] * gg[u][i].x; og[i].y += al[u] * gg[u][i].y;
                og[i].z += al[u] * gg[u][i].z; og[i].w += al[u] * gg[u][i].w;
            }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int a4 = lane + 64 * i;
        if (a4 < na4) {
            reinterpret_cast<float4 *>(part + (wave * 2 + 0) * A)[a4] = o[i];
            reinterpret_cast<float4 *>(part + (wave * 2 + 1) * A)[a4] = og[i];
        }
    }
    RSTAMP(4);
    __syncthreads();
    // ---- combine the waves' partials in ascending wave order: thread group k = tid / 256 -> (half, V | G); all of a
    // thread's partials are read together
    {
        const int k = tid >> 8, t = tid & 255;                 // k: 0 = v, 1 = sum Gc, 2 = s, 3 = sum Gs
        const int hf = k >> 1, vg = k & 1;
        if (t < na4) {
            float4 s4;
            if (hf == 0) {
                float4 x[RS_NWC];
#pragma unroll
                for (int w = 0; w < RS_NWC; ++w) x[w] = reinterpret_cast<const float4 *>(part + (w * 2 + vg) * A)[t];
                s4 = x[0];
#pragma unroll
                for (int w = 1; w < RS_NWC; ++w) { s4.x += x[w].x; s4.y += x[w].y; s4.z += x[w].z; s4.w += x[w].w; }
            } else {
                float4 x[RS_NWS];
#pragma unroll
                for (int w = 0; w < RS_NWS; ++w) x[w] = reinterpret_cast<const float4 *>(part + ((RS_NWC + w) * 2 + vg) * A)[t];
                s4 = x[0];
#pragma unroll
                for (int w = 1; w < RS_NWS; ++w) { s4.x += x[w].x; s4.y += x[w].y; s4.z += x[w].z; s4.w += x[w].w; }
            }
            reinterpret_cast<float4 *>(fin + k * A)[t] = s4;
        }
    }
    __syncthreads();
    RSTAMP(5);
    // ---- gate
    float dot = 0.f;
    float4 sv = z4, sw = z4;
    if (tid < na4) {
        sv = reinterpret_cast<const float4 *>(fin)[tid];
        const float4 sg = reinterpret_cast<const float4 *>(fin + A)[tid];
        sw = reinterpret_cast<const float4 *>(fin + 2 * A)[tid];
        const float4 gs = reinterpret_cast<const float4 *>(fin + 3 * A)[tid];
        const float4 zh = zh4, bc = bc4, bs = bs4, wg = wg4;
        const float zx = (zh.x + (sg.x + bc.x)) + (gs.x + bs.x);
        const float zy = (zh.y + (sg.y + bc.y)) + (gs.y + bs.y);
        const float zz = (zh.z + (sg.z + bc.z)) + (gs.z + bs.z);
        const float zw = (zh.w + (sg.w + bc.w)) + (gs.w + bs.w);
        dot = wg.x * isc_tanh(zx) + wg.y * isc_tanh(zy) + wg.z * isc_tanh(zz) + wg.w * isc_tanh(zw);
    }
    if (wave < 4) {
        dot = wave_sum(dot);
        if (lane == 0) redw[wave] = dot;
    }
    __syncthreads();
    if (tid < na4) {
        const float u = ((redw[0] + redw[1]) + (redw[2] + redw[3])) + b_gate;
        const float beta = isc_sigmoid(u);
        if (tid == 0 && a.beta) a.beta[(long long)b * a.beta_ld] = beta;
        float4 f;
        f.x = beta * sv.x + (1.0f - beta) * sw.x; f.y = beta * sv.y + (1.0f - beta) * sw.y;
        f.z = beta * sv.z + (1.0f - beta) * sw.z; f.w = beta * sv.w + (1.0f - beta) * sw.w;
        reinterpret_cast<float4 *>(a.f + (long long)b * A)[tid] = f;
        if (a.f_hi) store_planes4(a.f_hi, a.f_lo, b, 4 * tid, A, f);
    }
    RSTAMP(6);
#undef RSTAMP_KID
}

static bool rows_scan_ok(const isc_step_plan *p) {
    return p->att_e && p->words_e && p->gate_Gc && p->gate_Gs && p->A == p->E && p->A == p->W && p->A <= 512 &&
           (p->A & 3) == 0 && p->R >= 1 && p->Mw >= 1;
}

static int rows_scan_gate_launch(const RScanArgs &a, int rows, hipStream_t st) {
    if (!a.q[0] || !a.q[1] || !a.zh || !a.f || !a.w[0] || !a.w[1] || !a.b_c || !a.b_s || !a.w_g) return ISC_E_NULL;
    const size_t lds = ((size_t)((a.R[0] + 3) & ~3) + ((a.R[1] + 3) & ~3) + (size_t)(32 + 4) * a.A + 16) * sizeof(float);
    if (lds > 150000) return ISC_E_SHAPE;
    static std::atomic<bool> attr_set{false};
    if (!attr_set.load()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rows_scan_gate_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
        if (e != hipSuccess) return (int)e;
        attr_set.store(true);
    }
    hipLaunchKernelGGL(rows_scan_gate_kernel, dim3(rows), dim3(1024), lds, st, a);
    ISC_LAUNCH_CHECK();
    ++g_rows_launches;
    return ISC_OK;
}

static int rows_scan_gate(const isc_step_plan *p, int row_div, const int *skip, hipStream_t st) {
    RScanArgs a = {};
    a.row_div = row_div > 1 ? row_div : 1; a.skip = skip;
    a.P[0] = p->att_p; a.V[0] = p->att_e; a.G[0] = p->gate_Gc; a.q[0] = p->qa; a.w[0] = p->w_alpha_c; a.wb[0] = p->b_alpha_c;
    a.P[1] = p->words_p; a.V[1] = p->words_e; a.G[1] = p->gate_Gs; a.q[1] = p->qw; a.w[1] = p->w_alpha_s; a.wb[1] = p->b_alpha_s;
    a.q2 = p->label_w;
    a.ids = reinterpret_cast<const long long *>(p->words_ids); a.ids_ld = p->words_ids_ld;
    a.R[0] = p->R; a.R[1] = p->Mw; a.A = p->A;
    a.zh = p->z; a.b_c = p->b_gc; a.b_s = p->b_gs; a.w_g = p->w_gate; a.b_g = p->b_gate;
    a.f = p->f; a.alpha[0] = p->alpha_c; a.alpha[1] = p->alpha_s; a.beta = p->beta;
    a.alpha_ld[0] = p->alpha_c_ld; a.alpha_ld[1] = p->alpha_s_ld; a.beta_ld = p->beta_ld;
    return rows_scan_gate_launch(a, p->rows, st);
}

// isc_attn_scan_gate_fwd (attention.hip) for inference steps of up to g_rows_scan_max rows: the same kernel, one
// 1024-thread workgroup per row with the row's 290 KB in flight at once, against a 512-thread workgroup walking the
// regions (B = 128: 15.0 -> 10.9 us; one round of the chip up to 256 rows, slower beyond).  Needs the fused form
// (v and s not wanted on their own), <= 36 regions / 12 words, A = D <= 512.
#define ROWS_SCAN_MAX_ROWS 256
static std::atomic<int> g_rows_scan_max{ROWS_SCAN_MAX_ROWS};
extern "C" int isc_set_rows_scan_max(int rows) {
    if (rows < 0) return g_rows_scan_max.load();
    return g_rows_scan_max.exchange(rows);
}

// Regions up to which such a launch takes the row kernel.  Up to 36 every region's rows are in flight at once (one round
// trip); beyond, the kernel walks the regions 36 at a time (the reference encoder's 14 x 14 = 196-region grid: six trips)
// and still beats the 512-thread region walk of attn_scan_gate_kernel: greedy roll-outs at 196 regions, B = 128 2.33 ->
// 1.9-2.0 ms, B = 256 3.78 -> 3.36 ms, B = 64 equal (tools/r5_r196_scan.py, round 5; rounds 3-4 gated this at 36).
static std::atomic<int> g_rows_scan_regions{256};
extern "C" int isc_set_rows_scan_regions(int regions) {
    if (regions < 0) return g_rows_scan_regions.load();
    return g_rows_scan_regions.exchange(regions);
}

int rows_scan_gate_try(const isc_scan_gate_args *g, int rows, hipStream_t st, int *rc) {
    const isc_scan_problem &c = g->scan[0], &s = g->scan[1];
    if (rows > g_rows_scan_max.load()) return 0;
    if (c.out || s.out || c.out_hi || s.out_hi || c.q2 || c.row_ids) return 0;
    if (c.A != c.D || s.A != c.A || s.D != c.A || c.A > 512 || (c.A & 3) || c.R < 1 || s.R < 1) return 0;
    if (c.R > g_rows_scan_regions.load() || s.R > RS_NWS * RS_RPW) return 0;
    if (!c.P || !c.V || !c.q || !c.w || !s.P || !s.V || !s.q || !s.w || !g->G[0] || !g->G[1]) return 0;
    RScanArgs a = {};
    a.row_div = 1;
    a.skip = isc_stream_gate_(st);         // (isc_set_stream_gate: a batched search that has ended skips its remaining steps)
    a.P[0] = c.P; a.V[0] = c.V; a.G[0] = g->G[0]; a.q[0] = c.q; a.w[0] = c.w; a.wb[0] = c.w_bias;
    a.P[1] = s.P; a.V[1] = s.V; a.G[1] = g->G[1]; a.q[1] = s.q; a.w[1] = s.w; a.wb[1] = s.w_bias;
    a.q2 = s.q2;
    a.ids = reinterpret_cast<const long long *>(s.row_ids); a.ids_ld = s.row_ids_ld;
    a.R[0] = c.R; a.R[1] = s.R; a.A = c.A;
    a.zh = g->zh; a.b_c = g->b_gc; a.b_s = g->b_gs; a.w_g = g->w_gate; a.b_g = g->b_gate;
    a.f = g->f; a.alpha[0] = c.alpha_out; a.alpha[1] = s.alpha_out; a.beta = g->beta;
    a.alpha_ld[0] = c.alpha_ld; a.alpha_ld[1] = s.alpha_ld; a.beta_ld = g->beta_ld;
    a.f_hi = static_cast<_Float16 *>(g->f_hi); a.f_lo = static_cast<_Float16 *>(g->f_lo);
    *rc = rows_scan_gate_launch(a, rows, st);
    return 1;
}

// ------------------------------------------------------------------ the step
extern "C" int isc_rows_step_supported(const isc_step_plan *p) {
    if (!p) return 0;
    if (p->rows < 1 || p->rows > ROWS_MAX) return 0;
    if (!rows_scan_ok(p)) return 0;
    if (p->g1 || p->g2 || p->out_mask || p->apply_logsoftmax || p->pair_rows_c) return 0;   // training / teacher-forced forms: isc_step_fwd
    {   // the classifier's statistics tiles (<= 64 columns each) must fit isc_beam_select's register-resident tile lists
        // (64 lanes x 4 tiles): V <= 16384.  Larger vocabularies take the general kernels.
        const int tw = isc_rows_stats_tile(p->V);
        if (tw <= 0 || (p->V + tw - 1) / tw > 256) return 0;
    }
    if ((p->H & 3) || (p->E & 3) || (p->W & 3) || (p->A & 3)) return 0;
    // slices: att-LSTM H (+ W) + H, lang-LSTM E + H + H, each cut at 256
    const int s1 = (p->H + 255) / 256 * 2 + (p->tab ? 0 : (p->W + 255) / 256);
    const int s2 = (p->E + 255) / 256 + (p->H + 255) / 256 * 2;
    if (s1 > ROWS_MAX_SLICE || s2 > ROWS_MAX_SLICE || (p->H + 255) / 256 > ROWS_MAX_SLICE) return 0;
    return 1;
}

#define RET(x)                 \
    do {                       \
        int rc__ = (x);        \
        if (rc__) return rc__; \
    } while (0)

extern "C" int isc_rows_step_fwd(const isc_step_plan *p, const isc_rows_ext *x, void *stream) {
    if (!p || !x) return ISC_E_NULL;
    if (!isc_rows_step_supported(p)) return ISC_E_SHAPE;
    if (!p->pmax || !p->psum || !p->pidx) return ISC_E_NULL;
    hipStream_t st = (hipStream_t)stream;
    const int rows = p->rows, H = p->H, E = p->E, A = p->A, W = p->W, V = p->V;
    const int ld1 = H + E + W, ld2 = E + H;
    const int64_t *src = x->src_row;
    const int row_div = x->row_div > 1 ? x->row_div : 1;
    const int *skip = x->live_in;
    if (rows % row_div) return ISC_E_SHAPE;
    {   // att-LSTM over cat[h_lang_prev, fc, xt] (captioner.py:174-175); fc / label / biases are in pre1
        const float *As[3], *Ws[3];
        int lda[3], ldw[3], K[3], ind[3], n = 0;
        As[n] = p->h2_prev; lda[n] = H; Ws[n] = p->Wih1; ldw[n] = ld1; K[n] = H; ind[n] = 1; ++n;
        if (!p->tab) { As[n] = p->xt; lda[n] = W; Ws[n] = p->Wih1 + H + E; ldw[n] = ld1; K[n] = W; ind[n] = 0; ++n; }
        As[n] = p->h1_prev; lda[n] = H; Ws[n] = p->Whh1; ldw[n] = H; K[n] = H; ind[n] = 1; ++n;
        RLstmArgs::Fin fin = {};
        const isc_rollout_step *f = x->fin_prev;
        if (f) {        // the previous step's greedy finalize rides on this launch (isc_rows_ext.fin_prev)
            if (!p->tab || row_div != 1 || src || skip || rows > ROWS_FIN_MAX) return ISC_E_SHAPE;
            if (f->forced || f->sample_u || f->xt_next) return ISC_E_SHAPE;
            if (!f->part_max || !f->part_sum || !f->part_idx || !f->seq || !f->seq_logprobs || !f->seq_masks ||
                !f->unfinished || !f->alive || !x->fin_unfinished_out)
                return ISC_E_NULL;
            if (f->B != rows || f->T <= 0 || f->t < 0 || f->t + 1 >= f->T || f->n_tile <= 0) return ISC_E_SHAPE;
            fin.pmax = f->part_max; fin.psum = f->part_sum; fin.pidx = f->part_idx;
            fin.n_tile = f->n_tile; fin.T = f->T; fin.t = f->t; fin.eos = f->eos_id;
            fin.seq = reinterpret_cast<long long *>(f->seq); fin.raw = reinterpret_cast<long long *>(f->raw_tokens);
            fin.lp = f->seq_logprobs; fin.masks = f->seq_masks;
            fin.unf_in = f->unfinished; fin.unf_out = x->fin_unfinished_out; fin.alive = f->alive;
        }
        RET(rows_lstm(As, lda, Ws, ldw, K, ind, n, rows, H, src, p->c1_prev, p->h1, p->c1, nullptr, nullptr, p->pre1,
                      p->tab, p->tok, p->tok_stride, row_div, skip, st, f ? &fin : nullptr));
    }
    {   // projections of h_att: h2att (content), h2word (sentiment), the gate's h-term
        const float *Ws[3] = {p->W_h2att, p->W_h2word, p->W_gh}, *bs[3] = {p->b_h2att, p->b_h2word, p->b_gh};
        float *Cs[3] = {p->qa, p->qw, p->z};
        const int N[3] = {A, A, A}, ldc[3] = {A, A, A};
        RET(rows_linear3(p->h1, H, H, rows, Ws, bs, Cs, N, ldc, 3, skip, st));
    }
    RET(rows_scan_gate(p, row_div, skip, st));
    {   // lang-LSTM over cat[f, h_att] (captioner.py:180-181)
        const float *As[3] = {p->f, p->h1, p->h2_prev}, *Ws[3] = {p->Wih2, p->Wih2 + E, p->Whh2};
        const int lda[3] = {E, H, H}, ldw[3] = {ld2, ld2, H}, K[3] = {E, H, H}, ind[3] = {0, 0, 1};
        RET(rows_lstm(As, lda, Ws, ldw, K, ind, 3, rows, H, src, p->c2_prev, p->h2, p->c2, p->b_ih2, p->b_hh2, nullptr,
                      nullptr, nullptr, 0, 1, skip, st));
    }
    RET(rows_vocab(p->h2, H, p->W_cls, H, p->b_cls, rows, V, H, x->stats_tile, p->pmax, p->psum, p->pidx, p->logits,
                   p->ld_logits, x, st));
    return ISC_OK;
}

// Single kernels of the step behind their own entry points (tests, tools/rows_lab.py)
extern "C" int isc_rows_vocab_fwd(const float *h, int ldh, const float *W, int ldw, const float *bias, int M, int V,
                                  int K, float *part_max, float *part_sum, int32_t *part_idx, float *logits,
                                  int64_t ld_logits, const isc_rows_ext *ext, void *stream) {
    if (!ext) return ISC_E_NULL;
    return rows_vocab(h, ldh, W, ldw, bias, M, V, K, ext->stats_tile, part_max, part_sum, part_idx, logits, ld_logits,
                      ext, (hipStream_t)stream);
}
