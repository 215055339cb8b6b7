// compress_ws.hip -- CompressBlock attention of a SORTED PILLAR level in ONE launch, weights stationary.
//
// The same arithmetic as compress_fused.hip (ref: mssvt_backbone.py:351-383, MixedScaleAttention mssvt_utils.py:112-150 with
// nq = 1, seq-first) for the case the detector runs: pillar windows [1, 1, z] on a voxel list sorted by (b, x, y, z), a list
// capacity that cannot truncate (max_num_win1 >= the slab height), one head group of C / 16 heads of 16 channels.  There a
// window IS a run of consecutive voxel rows, consecutive windows are consecutive runs, and nothing has to be handed through
// memory: compress_fused.hip writes and re-reads the key tokens (N, C), the V rows (N, C), the scores and the projected
// queries between its three launches (308 MB of HBM traffic against ~60 MB of input + output) and walks every window's list
// twice with one dependent load per slot.
//
// One workgroup of C / 16 waves per CU; wave w keeps, as split-fp16 MFMA A-fragments (see ffn.hip, k_ffn_ws) in REGISTERS for
// the whole launch, rows [16 w, 16 w + 16) of pos_proj.2, Wk and Wv -- i.e. everything of HEAD w -- and reads its rows of
// Wq and Wo from LDS (used once per 16 windows).  A tile = 16 consecutive windows = one run of rows [r0, r1):
//   Q    the rows once, coalesced: channel-wise max per window through LDS integer atomics (order independent: exact),
//        q' = scale log2(e) (Wq q_tok + bq): wave w ends with head w of the 16 queries in registers          [2 barriers]
//   per 16 rows of the run:
//   S1   h = relu(pos_proj.0 [rel ; centre] + b): each wave its 16 channels, split, published as B fragments  [barrier]
//   S2   k_tok = xhat + relu(pos_proj.2 h + b): each wave its 16 channels, split, published                   [barrier]
//   S3   K and V of head w; score = q' . K (q' of the row's window by ds_bpermute); the softmax-weighted sum of V over
//        each window's rows as a SEGMENTED SCAN over the 16 row lanes (the rows of a window are adjacent lanes; DPP
//        row shifts; the running (max, sum, sum p V) of a window that continues in the next 16 rows is carried in
//        registers); a window's last row normalises, splits and publishes head w of the attention output
//   O    out = Wo o + bo for the 16 windows, each wave its 16 channels, whole rows of `out`                   [barrier]
// Deterministic: no floating-point atomics, fixed association inside a window.  Differences to compress_fused.hip:
// summation order of the softmax (scan tree instead of slot order) and 2^x instead of e^x -- rounding only.
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

#define CW_SCALE 2048.0f
#define CW_INV (1.0f / 2048.0f)
#define CW_LOG2E 1.4426950408889634f
#define MFMA_H(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16((av), (bv), acc, 0, 0, 0)
#define MFMA4(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av), (bv), acc, 0, 0, 0)
#define CW_MATS 5  // packed order: pos_proj.2, Wk, Wv (registers) | Wq, Wo (LDS)

struct CwArgs {
    int ns, num_voxels, z_magic;  // (z * z_magic) >> 16 == z / z_ws for every cell z < 64
    float qscale;  // attention scale x log2(e)
    const int *num_wins;                      // device
    const int *indices;                       // (N, 4) [b, z, y, x]
    const int *k_ind, *win_vstart, *win_cnt;  // K4 lists: only the run [vstart + min(list), + cnt) is taken from them
    const int *pair_win;                      // (N) window of every voxel, -1: in no list (cells above the window grid)
    float vsx, vsy, vsz, minx, miny, minz, wsx, wsy, wsz;
    const float *xhat;
    const float *Wp1, *bp1, *bp2, *bq, *bkv, *bo;
    float *out;
};

__device__ __forceinline__ int lane_pick4i(int g, int x, int y, int z, int w) {
    int r = w;
    r = g == 2 ? z : r;
    r = g == 1 ? y : r;
    r = g == 0 ? x : r;
    return r;
}
__device__ __forceinline__ float cw_centre(int idx, float cell, float lo) {
    return __fadd_rn(__fmul_rn(__fadd_rn((float)idx, 0.5f), cell), lo);  // ref with_coords, mssvt_backbone.py:132-137
}

// v -> (hi, lo): hi = fp16(v), lo = fp16((v - hi) 2^11), round toward zero (the arithmetic of ffn.hip)
__device__ __forceinline__ void cw_split4(const float v0, const float v1, const float v2, const float v3, h16x4 &hi, h16x4 &lo) {
    const fp16x2 a = __builtin_amdgcn_cvt_pkrtz(v0, v1), b = __builtin_amdgcn_cvt_pkrtz(v2, v3);
    const fp16x2 c = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)a[0], -CW_SCALE, v0 * CW_SCALE), __builtin_fmaf((float)a[1], -CW_SCALE, v1 * CW_SCALE));
    const fp16x2 d = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)b[0], -CW_SCALE, v2 * CW_SCALE), __builtin_fmaf((float)b[1], -CW_SCALE, v3 * CW_SCALE));
    hi = h16x4{(_Float16)a[0], (_Float16)a[1], (_Float16)b[0], (_Float16)b[1]};
    lo = h16x4{(_Float16)c[0], (_Float16)c[1], (_Float16)d[0], (_Float16)d[1]};
}
__device__ __forceinline__ h16x8 cw_cat(const h16x4 a, const h16x4 b) { return h16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// order-preserving float <-> int (signed compare): the channel-wise max goes through ds_max_i32
__device__ __forceinline__ int cw_key(float v) {
    const int b = __builtin_bit_cast(int, v);
    return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float cw_unkey(int k) { return __builtin_bit_cast(float, k >= 0 ? k : k ^ 0x7FFFFFFF); }

// (functions taking the value: __builtin_bit_cast applied directly to an ELEMENT of an ext_vector -- o[i] -- reads element 0)
__device__ __forceinline__ int cw_bits(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ float cw_float(int v) { return __builtin_bit_cast(float, v); }
#define CW_DPP_I(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
#define CW_DPP_F(v, ctrl) cw_float(__builtin_amdgcn_update_dpp(0, cw_bits(v), (ctrl), 0xF, 0xF, true))

// A-fragments of all waves, split once per parameter version: [matrix][wave][k step P][hi | lo][lane] x 16 bytes.  Lane
// (la, g) of wave w: row 16 w + la, k slot (g, j) <-> input channel 32 P + 16 (j / 4) + 4 g + j % 4 -- the accumulator layout
// of the producing product (ffn.hip, W2 fragments), so that no B operand is ever re-laid out
template <int C>
__global__ void __launch_bounds__(MSSVT_WAVE) k_cmp_ws_pack(const float *Wp2, const float *Wq, const float *Wkv, const float *Wo, h16x8 *packed) {
    constexpr int NW = C / 16, NP = C / 32;
    const int m = blockIdx.x / NW, wv = blockIdx.x % NW, lane = lane_id(), la = lane & 15, g = lane >> 4;
    const float *W = m == 0 ? Wp2 : m == 1 ? Wkv : m == 2 ? Wkv + (size_t)C * C : m == 3 ? Wq : Wo;
    h16x8 *dst = packed + ((size_t)(m * NW + wv) * NP * 2) * 64 + lane;
#pragma unroll
    for (int P = 0; P < NP; ++P) {
        const float *src = W + (size_t)(16 * wv + la) * C + 32 * P + 4 * g;
        const float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 16);
        h16x4 h0, l0, h1, l1;
        cw_split4(v0.x, v0.y, v0.z, v0.w, h0, l0);
        cw_split4(v1.x, v1.y, v1.z, v1.w, h1, l1);
        dst[(P * 2) * 64] = cw_cat(h0, h1);
        dst[(P * 2 + 1) * 64] = cw_cat(l0, l1);
    }
}

#ifdef CW_DEBUG
__device__ float *g_cw_dbg[3];  // q_tok (nw, C), q' (nw, C), scores (N, heads)
extern "C" int mssvt_debug_cmp_ws(float *qtok, float *qp, float *score) {
    float *h[3] = {qtok, qp, score};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cw_dbg), h, sizeof(h));
}
#endif

#ifdef CW_STAMPS  // developer instrumentation: shader clocks per phase, per wave of the first 8 workgroups
__device__ unsigned long long g_cw_stamps[8 * 8 * 16];
extern "C" int mssvt_debug_cmp_ws_stamps(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cw_stamps), sizeof(g_cw_stamps));
}
#define CSTAMP(k_) { const unsigned long long t_ = __builtin_readcyclecounter(); cs_acc[k_] += t_ - cs_t; cs_t = t_; }
#else
#define CSTAMP(k_)
#endif

template <int C>
__global__ void __launch_bounds__((C / 16) * MSSVT_WAVE, 1) k_cmp_ws(CwArgs a, const h16x8 *packed) {
#ifdef CW_STAMPS
    unsigned long long cs_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, cs_t = __builtin_readcyclecounter();
#endif
    constexpr int NW = C / 16, NP = C / 32, LPR = C / 4, RPW = MSSVT_WAVE / LPR, FR = NP * 2 * 64;  // FR: h16x8 per fragment set
    static_assert(NW * RPW == 16 && (LPR == 32 || LPR == 16 || LPR == 8), "16 rows per pass of the row-wise view");
    extern __shared__ float4 lds4[];
    h16x8 *wq_l = reinterpret_cast<h16x8 *>(lds4);  // [wave][P][hi | lo][lane]
    h16x8 *wo_l = wq_l + NW * FR;
    h16x8 *hfrag = wo_l + NW * FR;                  // B operands of the pos_proj.2 product
    h16x8 *kfrag = hfrag + FR;                      // ... of the K / V products
    h16x8 *ofrag = kfrag + FR;                      // ... of the Wo product
    // keys of the channel-wise max of the NEXT tile's windows, [16 windows][C]; channel c of window w sits at
    // w C + ((c + 4 w) mod C): the 16 windows of a fragment read start in different banks
    int *qmax = reinterpret_cast<int *>(ofrag + FR);
#define CW_QSLOT(w_, c_) ((w_) * C + (((c_) + 4 * (w_)) & (C - 1)))
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    const int r = wv * RPW + lane / LPR, q = lane % LPR;  // row-wise view: row of a 16-row pass, channels [4 q, 4 q + 4)
    const int nw = *a.num_wins, tiles = (nw + 15) >> 4;
    if (tiles <= 0) return;
    // ---- this workgroup's tiles: a CONTIGUOUS range holding ~1 / gridDim of the COST.  16 rows of a run cost ~5000 clocks, the
    // per-tile steps (Q, O, their barriers) ~2500 per 16 windows, and a tile holds 16 to 130 rows at 160k points (dealt
    // round-robin the busiest CU carries 1.5 x the mean; split by rows alone, the CUs of the sparse far field get 18
    // tiles instead of 8): cost before row r = 8 r + 5 pair_win[r] (windows are numbered in row order), range c starts
    // at the first tile boundary at or after cost c total / gridDim.  Found by one probe per thread + one scan of the
    // bracket: two dependent loads, no prefix sums.  (Any f(c) with f(0) = 0, f(grid) = tiles covers every tile.)
    int tile, tile_end;
    {
        int *ps = qmax;  // [probes | 2 results | 2 brackets]
        const int T = NW * MSSVT_WAVE, n = a.num_voxels;
        const int ri = (int)((long long)threadIdx.x * n / T);
        ps[threadIdx.x] = 8 * ri + 5 * max(a.pair_win[min(ri, n - 1)], 0);
        if (threadIdx.x < 2) {
            ps[T + threadIdx.x] = 0x7FFFFFFF;
            ps[T + 2 + threadIdx.x] = -1;
        }
        __syncthreads();
        const long long total = 8ll * n + 5ll * nw;
        long long tgt[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            tgt[e] = total * ((int)blockIdx.x + e) / (int)gridDim.x;
            if (ps[threadIdx.x] < tgt[e] && (threadIdx.x == T - 1 || ps[threadIdx.x + 1] >= tgt[e])) ps[T + 2 + e] = threadIdx.x;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int b = ps[T + 2 + e];
            if (b < 0) continue;  // (target at or below the first probe: boundary 0)
            const int lo = (int)((long long)b * n / T), hi = b + 1 < T ? (int)((long long)(b + 1) * n / T) + 1 : n;
            for (int row = lo + threadIdx.x; row < min(hi, n); row += T) {
                const int pw = a.pair_win[row];
                if (pw >= 0 && 8ll * row + 5ll * pw >= tgt[e]) atomicMin(ps + T + e, pw);
            }
        }
        __syncthreads();
        int ends[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int cc = (int)blockIdx.x + e, w = ps[T + e];
            ends[e] = cc <= 0 ? 0 : cc >= (int)gridDim.x ? tiles : ps[T + 2 + e] < 0 ? 0 : w == 0x7FFFFFFF ? tiles : min((w + 15) >> 4, tiles);
        }
        tile = ends[0];
        tile_end = ends[1];
        __syncthreads();  // (the scratch becomes the max tile)
    }
    if (tile >= tile_end) return;

    // ---- the run of rows of a tile: [first row of its first window, last row of its last window] -----------------------
    // lanes 0-31: the K4 list of the first window, lanes 32-63: of the last one (k_ind is -1 beyond the list): three
    // independent loads, issued a tile ahead and reduced when needed
    int run_v = 0, run_c = 0, run_s = 0;
#define CW_RUN_ISSUE(tile_)                                                                                   \
    {                                                                                                         \
        const int t_ = min((tile_), tiles - 1), w0_ = t_ * 16;                                                \
        const int w_ = lane < 32 ? w0_ : min(w0_ + 15, nw - 1);                                               \
        run_c = a.win_cnt[w_];                                                                                \
        run_s = a.win_vstart[w_];                                                                             \
        run_v = (lane & 31) < a.ns ? a.k_ind[(size_t)w_ * a.ns + (lane & 31)] : -1;                           \
    }
#define CW_RUN_TAKE(r0_, r1_)                                                                                 \
    {                                                                                                         \
        int v_ = run_v < 0 ? 0x7FFFFFFF : run_v;                                                              \
        _Pragma("unroll") for (int off_ = 1; off_ < 32; off_ <<= 1) v_ = min(v_, __shfl_xor(v_, off_));       \
        r0_ = __builtin_amdgcn_readlane(run_s + v_, 0);                                                       \
        r1_ = __builtin_amdgcn_readlane(run_s + v_ + run_c, 32);                                              \
        if (r1_ < r0_ || r0_ < 0 || r1_ > a.num_voxels) r1_ = r0_ = 0; /* (only after a table overflow) */     \
    }
    // one pass of the channel-wise max: rows [rb_, rb_ + 32) of a run ending at re_ -> qmax (windows of the tile at w0_)
#define CW_MAX_LOAD(rb_, re_)                                                                                 \
    {                                                                                                         \
        const int ra_ = (rb_) + r, rc_ = (rb_) + 16 + r;                                                      \
        mxa = mxb = make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
        mpa = mpb = -1;                                                                                       \
        if (ra_ < (re_)) {                                                                                    \
            mxa = *reinterpret_cast<const float4 *>(a.xhat + (size_t)ra_ * C + 4 * q);                        \
            mpa = a.pair_win[ra_];                                                                            \
        }                                                                                                     \
        if (rc_ < (re_)) {                                                                                    \
            mxb = *reinterpret_cast<const float4 *>(a.xhat + (size_t)rc_ * C + 4 * q);                        \
            mpb = a.pair_win[rc_];                                                                            \
        }                                                                                                     \
    }
#define CW_MAX_PUT(w0_, nwt_)                                                                                 \
    {                                                                                                         \
        if ((unsigned int)(mpa - (w0_)) < (unsigned int)(nwt_)) {                                             \
            const int w_ = mpa - (w0_);                                                                       \
            atomicMax(qmax + CW_QSLOT(w_, 4 * q), cw_key(mxa.x)); atomicMax(qmax + CW_QSLOT(w_, 4 * q + 1), cw_key(mxa.y)); \
            atomicMax(qmax + CW_QSLOT(w_, 4 * q + 2), cw_key(mxa.z)); atomicMax(qmax + CW_QSLOT(w_, 4 * q + 3), cw_key(mxa.w)); \
        }                                                                                                     \
        if ((unsigned int)(mpb - (w0_)) < (unsigned int)(nwt_)) {                                             \
            const int w_ = mpb - (w0_);                                                                       \
            atomicMax(qmax + CW_QSLOT(w_, 4 * q), cw_key(mxb.x)); atomicMax(qmax + CW_QSLOT(w_, 4 * q + 1), cw_key(mxb.y)); \
            atomicMax(qmax + CW_QSLOT(w_, 4 * q + 2), cw_key(mxb.z)); atomicMax(qmax + CW_QSLOT(w_, 4 * q + 3), cw_key(mxb.w)); \
        }                                                                                                     \
    }
    // initial keys of a tile's windows: 0 (= key(0.0f): the zero padding takes part, ref :370) unless the list is full
#define CW_MAX_INIT(w0_, nwt_)                                                                                \
    {                                                                                                         \
        const int w_ = threadIdx.x / LPR; /* NW * 64 / LPR = 16 windows */                                    \
        const int cnt_ = a.win_cnt[min((w0_) + w_, nw - 1)];                                                  \
        const int init_ = w_ < (nwt_) && cnt_ >= a.ns ? cw_key(-INFINITY) : 0;                                \
        *reinterpret_cast<int4 *>(qmax + w_ * C + 4 * q) = make_int4(init_, init_, init_, init_);             \
    }
    float4 mxa, mxb;
    int mpa, mpb;

    int r0, r1, nr0 = 0, nr1 = 0;
    CW_RUN_ISSUE(tile)
    CW_MAX_INIT(tile * 16, min(16, nw - tile * 16))
    // ---- weights ---------------------------------------------------------------------------------------------------
    h16x8 Wph[NP], Wpl[NP], Wkh[NP], Wkl[NP], Wvh[NP], Wvl[NP];
    {
        const h16x8 *src = packed + (size_t)wv * FR + lane;
#pragma unroll
        for (int P = 0; P < NP; ++P) {
            Wph[P] = src[(P * 2) * 64];
            Wpl[P] = src[(P * 2 + 1) * 64];
            Wkh[P] = src[(size_t)NW * FR + (P * 2) * 64];
            Wkl[P] = src[(size_t)NW * FR + (P * 2 + 1) * 64];
            Wvh[P] = src[(size_t)2 * NW * FR + (P * 2) * 64];
            Wvl[P] = src[(size_t)2 * NW * FR + (P * 2 + 1) * 64];
        }
        // Wq | Wo fragments of every wave -> LDS, lane-linear (2 NW FR x 16 bytes)
        const h16x8 *s2 = packed + (size_t)3 * NW * FR;
        constexpr int UN = 8;
        for (int e0 = threadIdx.x; e0 < 2 * NW * FR; e0 += NW * MSSVT_WAVE * UN) {
            h16x8 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + u * NW * MSSVT_WAVE;
                v[u] = s2[e < 2 * NW * FR ? e : 0];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int e = e0 + u * NW * MSSVT_WAVE;
                if (e < 2 * NW * FR) wq_l[e] = v[u];
            }
        }
    }
    const int ch = 16 * wv + 4 * g;  // MFMA view: this lane's 4 output channels
    const float4 bp2 = *reinterpret_cast<const float4 *>(a.bp2 + ch), bq = *reinterpret_cast<const float4 *>(a.bq + ch),
                 bk = *reinterpret_cast<const float4 *>(a.bkv + ch), bv = *reinterpret_cast<const float4 *>(a.bkv + C + ch),
                 bo = *reinterpret_cast<const float4 *>(a.bo + ch);
    // pos_proj.0 as a K = 8 product: inputs d = 4 s + g of step s = (rel.x, rel.y, rel.z, c.x | c.y, c.z, 1, 0); A operand
    // of this lane: row = channel 16 w + la, extended by the bias
    float w1a0, w1a1;
    {
        const float *wr = a.Wp1 + (size_t)(16 * wv + la) * 6;
        w1a0 = wr[g];
        w1a1 = lane_pick4(g, wr[4], wr[5], a.bp1[16 * wv + la], 0.f);
    }
    const int ks = wv >> 1, hs = wv & 1;  // this wave's 16 channels inside the fragments: k step, half of the 8 slots
    // per-lane constants of the positional inputs (k slot g): (rel.x, rel.y, rel.z, c.x | c.y, c.z, 1, 0)
    const float g_cv = lane_pick4(g, a.vsx, a.vsy, a.vsz, 0.f), g_lo0 = lane_pick4(g, a.minx, a.miny, a.minz, a.minx),
                g_cw0 = lane_pick4(g, a.wsx, a.wsy, a.wsz, a.wsx), g_cw1 = lane_pick4(g, a.wsy, a.wsz, 0.f, 0.f),
                g_lo1 = lane_pick4(g, a.miny, a.minz, 0.f, 0.f);

    // ---- the first tile's channel-wise max (every later tile's is taken under the tile before it) -----------------------
    CW_RUN_TAKE(r0, r1)
    CW_RUN_ISSUE(tile + 1)
    __syncthreads();
    CSTAMP(0)  // weights, partition, first run
    for (int rb = r0; rb < r1; rb += 32) {
        CW_MAX_LOAD(rb, r1)
        CW_MAX_PUT(tile * 16, min(16, nw - tile * 16))
    }
    __syncthreads();
    CSTAMP(1)  // first tile's max

    // rows of the MFMA view, requested 16 rows ahead: window, next row's window, cell, this wave's 16 channels of xhat
#define CW_ROW_LOAD(rb_, pw_, pn_, vi_, xs_)                                                                  \
    {                                                                                                         \
        const int row_ = max(min((rb_) + la, r1 - 1), 0);                                                     \
        pw_ = a.pair_win[row_];                                                                               \
        pn_ = a.pair_win[min(row_ + 1, a.num_voxels - 1)];                                                    \
        vi_ = reinterpret_cast<const int4 *>(a.indices)[row_];                                                \
        xs_ = *reinterpret_cast<const float4 *>(a.xhat + (size_t)row_ * C + ch);                              \
    }
    for (;;) {
        const int w0 = tile * 16, nwt = min(16, nw - w0);
        const int tile_n = tile + 1;
        const bool has_n = tile_n < tile_end;
        const int w0n = min(tile_n, tiles - 1) * 16, nwtn = has_n ? min(16, nw - w0n) : 0;
        // ---- Q: q' = scale log2(e) (Wq q_tok + bq): head w of the 16 queries, lane (la = window, g): channels 16 w + 4 g + i
        f32x4 qp;
        {
            f32x4 m = f32x4{bq.x, bq.y, bq.z, bq.w}, l = f32x4{0.f, 0.f, 0.f, 0.f}, k = l;
            const h16x8 *wf = wq_l + (size_t)wv * FR + lane;
#pragma unroll
            for (int P = 0; P < NP; ++P) {
                const int4 k0 = *reinterpret_cast<const int4 *>(qmax + CW_QSLOT(la, 32 * P + 4 * g)),
                           k1 = *reinterpret_cast<const int4 *>(qmax + CW_QSLOT(la, 32 * P + 16 + 4 * g));
                h16x4 h0, l0, h1, l1;
                cw_split4(cw_unkey(k0.x), cw_unkey(k0.y), cw_unkey(k0.z), cw_unkey(k0.w), h0, l0);
                cw_split4(cw_unkey(k1.x), cw_unkey(k1.y), cw_unkey(k1.z), cw_unkey(k1.w), h1, l1);
                const h16x8 bh = cw_cat(h0, h1), bl = cw_cat(l0, l1);
                const h16x8 ah = wf[(P * 2) * 64], al = wf[(P * 2 + 1) * 64];
                MFMA_H(m, ah, bh);
                MFMA_H(l, ah, bl);
                MFMA_H(k, al, bh);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) qp[i] = __builtin_fmaf(l[i] + k[i], CW_INV, m[i]) * a.qscale;
#ifdef CW_DEBUG
            if (la < nwt && g_cw_dbg[1])
                for (int i = 0; i < 4; ++i) g_cw_dbg[1][(size_t)(w0 + la) * C + ch + i] = qp[i];
#endif
        }
        // the next tile's run (requested a tile ago); the one after it is requested now
        CW_RUN_TAKE(nr0, nr1)
        if (!has_n) nr0 = nr1 = 0;
        CW_RUN_ISSUE(tile_n + 1)
        int pw_n, pn_n;
        int4 vi_n;
        float4 xs_n;
        CW_ROW_LOAD(r0, pw_n, pn_n, vi_n, xs_n)
        CSTAMP(2)  // Q product
        __syncthreads();  // this tile's max keys are consumed: the tile becomes the next tile's
        CSTAMP(3)
        CW_MAX_INIT(w0n, nwtn)
#ifdef CW_STAMPS
        cs_acc[14] += 1;
#endif
        int np = nr0;     // next tile's rows whose max is taken so far

        // ---- the run, 16 rows at a time ------------------------------------------------------------------------------
        float cm = 0.f, cs = 0.f;  // carried (max, sum, sum p V) of the window open at the end of the previous 16 rows
        f32x2 co01 = f32x2{0.f, 0.f}, co23 = co01;
        int cseg = -1;
        bool first = true;
        for (int rb = r0; rb < r1; rb += 16) {
            const int row = min(rb + la, r1 - 1);
            const bool rlive = rb + la < r1;
            const int pw = pw_n, pnext = pn_n;
            const int4 vi = vi_n;
            const float4 xs = xs_n;
            CW_ROW_LOAD(rb + 16, pw_n, pn_n, vi_n, xs_n)
            CW_MAX_LOAD(np, nr1)
            const bool rvalid = rlive && (unsigned int)(pw - w0) < (unsigned int)nwt;
            const bool wend = row + 1 >= r1 || pnext != pw;  // the window's last row
            // S1: h = relu(pos_proj.0 [rel ; centre] + b), this wave's 16 channels.  Pillar windows: the window's cell is
            // the voxel's (x, y) and z / z_ws.  A lane forms only the components it feeds to the product (k slot g): the
            // integer inputs are picked first, then one voxel centre and two window centres -- the same operations per
            // component as cw_centre on all six (ref with_coords), a third of the instructions
            {
                const int wz = (vi.y * a.z_magic) >> 16;
                const int iv = lane_pick4i(g, vi.w, vi.z, vi.y, 0), iw0 = lane_pick4i(g, vi.w, vi.z, wz, vi.w), iw1 = lane_pick4i(g, vi.z, wz, 0, 0);
                const float wc0 = cw_centre(iw0, g_cw0, g_lo0), wc1 = cw_centre(iw1, g_cw1, g_lo1);
                const float rel = cw_centre(iv, g_cv, g_lo0) - wc0;  // NOT masked in the CompressBlock (ref :372)
                const float in0 = g < 3 ? rel : wc0, in1 = g < 2 ? wc1 : g == 2 ? 1.0f : 0.0f;
                f32x4 p = f32x4{0.f, 0.f, 0.f, 0.f};
                MFMA4(p, w1a0, in0);
                MFMA4(p, w1a1, in1);
                h16x4 hi, lo;
                cw_split4(fmaxf(p[0], 0.f), fmaxf(p[1], 0.f), fmaxf(p[2], 0.f), fmaxf(p[3], 0.f), hi, lo);
                reinterpret_cast<h16x4 *>(hfrag + (ks * 2) * 64 + lane)[hs] = hi;
                reinterpret_cast<h16x4 *>(hfrag + (ks * 2 + 1) * 64 + lane)[hs] = lo;
            }
            CSTAMP(4)  // S1 (+ prefetch issue)
            __syncthreads();
            CSTAMP(5)
            // S2: k_tok = xhat + relu(pos_proj.2 h + b), this wave's 16 channels
            {
                f32x4 m = f32x4{bp2.x, bp2.y, bp2.z, bp2.w}, l = f32x4{0.f, 0.f, 0.f, 0.f}, k = l;
#pragma unroll
                for (int P = 0; P < NP; ++P) {
                    const h16x8 bh = hfrag[(P * 2) * 64 + lane], bl = hfrag[(P * 2 + 1) * 64 + lane];
                    MFMA_H(m, Wph[P], bh);
                    MFMA_H(l, Wph[P], bl);
                    MFMA_H(k, Wpl[P], bh);
                }
                h16x4 hi, lo;
                cw_split4(xs.x + fmaxf(__builtin_fmaf(l[0] + k[0], CW_INV, m[0]), 0.f), xs.y + fmaxf(__builtin_fmaf(l[1] + k[1], CW_INV, m[1]), 0.f),
                          xs.z + fmaxf(__builtin_fmaf(l[2] + k[2], CW_INV, m[2]), 0.f), xs.w + fmaxf(__builtin_fmaf(l[3] + k[3], CW_INV, m[3]), 0.f), hi, lo);
                reinterpret_cast<h16x4 *>(kfrag + (ks * 2) * 64 + lane)[hs] = hi;
                reinterpret_cast<h16x4 *>(kfrag + (ks * 2 + 1) * 64 + lane)[hs] = lo;
            }
            CSTAMP(6)  // S2
            __syncthreads();
            CSTAMP(7)
            // S3: K, V of head w; scores; segmented softmax-weighted sum over the row lanes
            {
                f32x4 km = f32x4{bk.x, bk.y, bk.z, bk.w}, kl = f32x4{0.f, 0.f, 0.f, 0.f}, kk = kl;
                f32x4 vm = f32x4{bv.x, bv.y, bv.z, bv.w}, vl = kl, vk = kl;
#pragma unroll
                for (int P = 0; P < NP; ++P) {
                    const h16x8 bh = kfrag[(P * 2) * 64 + lane], bl = kfrag[(P * 2 + 1) * 64 + lane];
                    MFMA_H(km, Wkh[P], bh);
                    MFMA_H(vm, Wvh[P], bh);
                    MFMA_H(kl, Wkh[P], bl);
                    MFMA_H(vl, Wvh[P], bl);
                    MFMA_H(kk, Wkl[P], bh);
                    MFMA_H(vk, Wvl[P], bh);
                }
                const int widx = rvalid ? pw - w0 : 0;
                const int src = 4 * (16 * g + widx);
                float sc = 0.f;
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float qi = cw_float(__builtin_amdgcn_ds_bpermute(src, cw_bits(qp[i])));
                    sc = __builtin_fmaf(__builtin_fmaf(kl[i] + kk[i], CW_INV, km[i]), qi, sc);
                    o[i] = __builtin_fmaf(vl[i] + vk[i], CW_INV, vm[i]);
                }
                sc += lane_xor16(sc);
                sc += lane_xor32(sc);
                CSTAMP(8)  // K, V, scores
#ifdef CW_DEBUG
                if (rvalid && g == 0 && g_cw_dbg[2]) g_cw_dbg[2][(size_t)row * NW + wv] = sc;
#endif
                float m_ = sc, s_ = 1.0f;
                const int seg = rvalid ? pw : -1 - la;  // rows outside the tile's windows: segments of their own, never published
                // (max, sum, sum p V) of two adjacent pieces of one window
#define CW_COMBINE(mg_, pm_, ps_, po01_, po23_)                                                          \
    {                                                                                                   \
        const float nm_ = fmaxf(pm_, m_);                                                               \
        /* no merge: weights (0, 1) -- every lane holds finite values, 0 * x = 0 */                     \
        const float ea_ = (mg_) ? __builtin_amdgcn_exp2f(pm_ - nm_) : 0.f, eb_ = (mg_) ? __builtin_amdgcn_exp2f(m_ - nm_) : 1.f; \
        s_ = __builtin_fmaf(ps_, ea_, s_ * eb_);                                                        \
        o01 = __builtin_elementwise_fma(po01_, f32x2{ea_, ea_}, o01 * f32x2{eb_, eb_});                 \
        o23 = __builtin_elementwise_fma(po23_, f32x2{ea_, ea_}, o23 * f32x2{eb_, eb_});                 \
        m_ = (mg_) ? nm_ : m_;                                                                          \
    }
                f32x2 o01 = f32x2{o[0], o[1]}, o23 = f32x2{o[2], o[3]};
                // the piece carried in from the previous 16 rows joins row lane 0
                if (!first) {  // (wave-uniform)
                    const bool mg0 = la == 0 && cseg == seg;
                    CW_COMBINE(mg0, cm, cs, co01, co23)
                }
                // Hillis-Steele over the 16 row lanes; a step whose distance no window of these rows reaches is skipped
                // (every DPP with all lanes active: a DPP source lane must be active)
                const int pg1 = CW_DPP_I(seg, 0x111), pg2 = CW_DPP_I(seg, 0x112), pg4 = CW_DPP_I(seg, 0x114), pg8 = CW_DPP_I(seg, 0x118);
                const bool mg1 = (la >= 1) & (pg1 == seg), mg2 = (la >= 2) & (pg2 == seg), mg4 = (la >= 4) & (pg4 == seg),
                           mg8 = (la >= 8) & (pg8 == seg);
#define CW_STEP(ctrl_, mg_)                                                                             \
    {                                                                                                   \
        const float pm_ = CW_DPP_F(m_, ctrl_), ps_ = CW_DPP_F(s_, ctrl_);                               \
        const float a0_ = o01[0], a1_ = o01[1], a2_ = o23[0], a3_ = o23[1];                             \
        const f32x2 p01_ = f32x2{CW_DPP_F(a0_, ctrl_), CW_DPP_F(a1_, ctrl_)}, p23_ = f32x2{CW_DPP_F(a2_, ctrl_), CW_DPP_F(a3_, ctrl_)}; \
        CW_COMBINE(mg_, pm_, ps_, p01_, p23_)                                                           \
    }
                if (__ballot(mg1) != 0ull) {
                    CW_STEP(0x111, mg1)
                    if (__ballot(mg2) != 0ull) {
                        CW_STEP(0x112, mg2)
                        if (__ballot(mg4) != 0ull) {
                            CW_STEP(0x114, mg4)
                            if (__ballot(mg8) != 0ull) CW_STEP(0x118, mg8)
                        }
                    }
                }
#undef CW_STEP
#undef CW_COMBINE
                if (rvalid && wend) {  // head w of the window's attention output, normalised -> B fragment of the Wo product
                    const float inv = 1.0f / s_;
                    h16x4 hi, lo;
                    cw_split4(o01[0] * inv, o01[1] * inv, o23[0] * inv, o23[1] * inv, hi, lo);
                    reinterpret_cast<h16x4 *>(ofrag + (ks * 2) * 64 + 16 * g + widx)[hs] = hi;
                    reinterpret_cast<h16x4 *>(ofrag + (ks * 2 + 1) * 64 + 16 * g + widx)[hs] = lo;
                }
                // carry: row lane 15's piece to row lane 0 of the next 16 rows (row_ror:1)
                cm = CW_DPP_F(m_, 0x121);
                cs = CW_DPP_F(s_, 0x121);
                {
                    const float a0_ = o01[0], a1_ = o01[1], a2_ = o23[0], a3_ = o23[1];
                    co01 = f32x2{CW_DPP_F(a0_, 0x121), CW_DPP_F(a1_, 0x121)};
                    co23 = f32x2{CW_DPP_F(a2_, 0x121), CW_DPP_F(a3_, 0x121)};
                }
                cseg = CW_DPP_I(seg, 0x121);
                first = false;
            }
            CSTAMP(9)  // scan, publish
            // the next tile's max, 32 rows per 16 rows of this tile (requested at the top of the iteration)
            CW_MAX_PUT(w0n, nwtn)
            np += 32;
            CSTAMP(10)  // atomics
#ifdef CW_STAMPS
            cs_acc[15] += 1;
#endif
        }
        if (r1 <= r0) __syncthreads();  // (no iteration ran: the keys' initial values must be in place)
        for (; np < nr1; np += 32) {    // what is left of the next tile's rows (its run is longer than twice this one)
            CW_MAX_LOAD(np, nr1)
            CW_MAX_PUT(w0n, nwtn)
        }
        CSTAMP(11)  // rest of the next tile's max
        __syncthreads();
        CSTAMP(12)
        // ---- O: out = Wo o + bo, this wave's 16 channels of the 16 windows ------------------------------------------------
        {
            f32x4 m = f32x4{bo.x, bo.y, bo.z, bo.w}, l = f32x4{0.f, 0.f, 0.f, 0.f}, k = l;
            const h16x8 *wf = wo_l + (size_t)wv * FR + lane;
#pragma unroll
            for (int P = 0; P < NP; ++P) {
                const h16x8 bh = ofrag[(P * 2) * 64 + lane], bl = ofrag[(P * 2 + 1) * 64 + lane];
                const h16x8 ah = wf[(P * 2) * 64], al = wf[(P * 2 + 1) * 64];
                MFMA_H(m, ah, bh);
                MFMA_H(l, ah, bl);
                MFMA_H(k, al, bh);
            }
            if (la < nwt)
                *reinterpret_cast<float4 *>(a.out + (size_t)(w0 + la) * C + ch) =
                    make_float4(__builtin_fmaf(l[0] + k[0], CW_INV, m[0]), __builtin_fmaf(l[1] + k[1], CW_INV, m[1]),
                                __builtin_fmaf(l[2] + k[2], CW_INV, m[2]), __builtin_fmaf(l[3] + k[3], CW_INV, m[3]));
        }
        CSTAMP(13)  // O product
        if (!has_n) break;
        tile = tile_n;
        r0 = nr0;
        r1 = nr1;
    }
#ifdef CW_STAMPS
    if (lane == 0 && blockIdx.x % 32 == 0) {
        for (int k = 0; k < 16; ++k) g_cw_stamps[((blockIdx.x / 32) * 8 + wv) * 16 + k] = cs_acc[k];
    }
#endif
#undef CW_QSLOT
#undef CW_RUN_ISSUE
#undef CW_RUN_TAKE
#undef CW_MAX_LOAD
#undef CW_MAX_PUT
#undef CW_MAX_INIT
#undef CW_ROW_LOAD
}

extern "C" long long mssvt_compress_ws_packed_bytes(int C) {
    if (C != 128) return 0;
    return (long long)CW_MATS * (C / 16) * (C / 32) * 2 * 64 * 16;
}

// pos_proj.2 (C, C), to_q (C, C), to_kv (2 C, C), proj (C, C) -> the A fragments k_cmp_ws keeps in registers / LDS
extern "C" int mssvt_compress_ws_pack(int C, const float *Wpos2, const float *Wq, const float *Wkv, const float *Wo, void *packed,
                                      void *stream) {
    if (!Wpos2 || !Wq || !Wkv || !Wo || !packed) return MSSVT_E_BADARG;
    if (C != 128) return MSSVT_E_TOOLARGE;
    k_cmp_ws_pack<128><<<CW_MATS * (128 / 16), MSSVT_WAVE, 0, (hipStream_t)stream>>>(Wpos2, Wq, Wkv, Wo, reinterpret_cast<h16x8 *>(packed));
    return mssvt_launch_status();
}

// CompressBlock attention (everything of mssvt_compress_fused) in one launch.  Preconditions (the caller's, fused.py
// _compress_ws_ok): a level set up as SORTED (mssvt_level_setup_sorted: windows numbered in row order), pillar windows
// x_ws = y_ws = 1 with max_num_win1 >= z_ws and <= 32 (no list is truncated: a window is one run of rows), one head group,
// head_dim 16, C = 128, operands inside the fp16 range (fused._compress_f16_ok).  MSSVT_E_TOOLARGE for other shapes.
extern "C" int mssvt_compress_ws(int C, int head_dim, float scale, int z_ws, int max_num_win1, int num_voxels,
                                 const int *num_wins_dev, int win_capacity, const int *indices, const int *k_ind, const int *win_vstart,
                                 const int *win_cnt, const int *pair_win, const float *host_voxel_size3,
                                 const float *host_range_min3, const float *host_win_size3, const float *xhat, const float *Wpos1,
                                 const float *bpos1, const float *bpos2, const float *bq, const float *bkv, const float *bo,
                                 const void *packed, float *out, void *stream) {
    if (!num_wins_dev || !indices || !k_ind || !win_vstart || !win_cnt || !pair_win || !host_voxel_size3 ||
        !host_range_min3 || !host_win_size3 || !xhat || !Wpos1 || !bpos1 || !bpos2 || !bq || !bkv || !bo || !packed || !out ||
        max_num_win1 <= 0 || num_voxels < 0 || win_capacity <= 0 || z_ws <= 0)
        return MSSVT_E_BADARG;
    if (C != 128 || head_dim != 16 || max_num_win1 > 32 || z_ws > max_num_win1) return MSSVT_E_TOOLARGE;
    if (num_voxels == 0) return MSSVT_OK;
    CwArgs a;
    a.ns = max_num_win1; a.num_voxels = num_voxels; a.z_magic = 65536 / z_ws + 1; a.qscale = scale * CW_LOG2E;
    a.num_wins = num_wins_dev; a.indices = indices;
    a.k_ind = k_ind; a.win_vstart = win_vstart; a.win_cnt = win_cnt; a.pair_win = pair_win;
    a.vsx = host_voxel_size3[0]; a.vsy = host_voxel_size3[1]; a.vsz = host_voxel_size3[2];
    a.minx = host_range_min3[0]; a.miny = host_range_min3[1]; a.minz = host_range_min3[2];
    a.wsx = host_win_size3[0]; a.wsy = host_win_size3[1]; a.wsz = host_win_size3[2];
    a.xhat = xhat; a.Wp1 = Wpos1; a.bp1 = bpos1; a.bp2 = bpos2; a.bq = bq; a.bkv = bkv; a.bo = bo; a.out = out;
    constexpr int CC = 128, NW = CC / 16, FR = (CC / 32) * 2 * 64;
    const size_t lds = (size_t)(2 * NW * FR + 3 * FR) * 16 + (size_t)16 * CC * 4;
    static_assert((size_t)(2 * (128 / 16) * ((128 / 32) * 2 * 64) + 3 * ((128 / 32) * 2 * 64)) * 16 + 16 * 128 * 4 <= 160 * 1024, "LDS budget");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_cmp_ws<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int tiles = (win_capacity + 15) / 16;
    const int grid = tiles < cus ? tiles : cus;
    k_cmp_ws<CC><<<grid, NW * MSSVT_WAVE, lds, (hipStream_t)stream>>>(a, reinterpret_cast<const h16x8 *>(packed));
    return mssvt_launch_status();
}
