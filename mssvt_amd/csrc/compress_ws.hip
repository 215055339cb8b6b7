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
// Wq and Wo from LDS (used once per 16 windows).  The workgroup owns a CHUNK: consecutive windows = consecutive rows holding
// 1 / gridDim of the cost (rows and windows weighed 8 : 5), found by 4096 probes of pair_win.  The chunk is walked as a
// STREAM of 16-row pieces, whatever the windows; windows are taken in GROUPS of 16 for the two window-side products:
//   per piece of 16 rows:
//   S1   h = relu(pos_proj.0 [rel ; centre] + b): each wave its 16 channels, split, published as B fragments  [barrier]
//   S2   k_tok = xhat + relu(pos_proj.2 h + b): each wave its 16 channels, split, published                   [barrier]
//   S3   K and V of head w; score = q' . K (q' of the row's window by ds_bpermute from the lane that holds it); the
//        softmax-weighted sum of V over each window's rows as a LEFT FOLD over the 16 row lanes (the rows of a window are
//        adjacent lanes; DPP row_shr:1; the running (max, sum, sum p V) of a window that continues in the next piece is
//        carried in registers); a window's last row normalises, splits and publishes head w of the attention output
//   per group of 16 windows, switched in flight:
//   max  the channel-wise max of the NEXT group's rows (the query tokens, ref :370), 32 rows per piece of the stream,
//        coalesced, through LDS integer atomics on order-preserving keys (order independent: exact)
//   Q    when the stream reaches the group: q' = scale log2(e) (Wq q_tok + bq), wave w ends with head w of the 16 queries
//        in registers (the queries of two groups are held: a piece straddles at most two)               [2 barriers]
//   O    when the group's last window ends: out = Wo o + bo, each wave its 16 channels of the 16 output rows   [barrier]
// Deterministic, and independent of the chunking: no floating-point atomics, and the association inside a window (left fold in
// row order, the carried piece as its prefix) does not depend on where the pieces or the chunks are cut -- a scene's rows are
// bit-identical whatever it shares a batch with.  Differences to compress_fused.hip: summation order of the softmax (row
// order instead of list-slot order), 2^x instead of e^x, rcp instead of a division -- rounding only.
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
    const int *win_cnt;                       // voxels of every window (a list of max_num_win1 entries is not zero padded)
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
    constexpr int NW = C / 16, NP = C / 32, LPR = C / 4, RPW = MSSVT_WAVE / LPR, FR = NP * 2 * 64, T = NW * MSSVT_WAVE;
    static_assert(NW * RPW == 16 && (LPR == 32 || LPR == 16 || LPR == 8), "16 rows per pass of the row-wise view");
    extern __shared__ float4 lds4[];
    h16x8 *wq_l = reinterpret_cast<h16x8 *>(lds4);  // [wave][P][hi | lo][lane]
    h16x8 *wo_l = wq_l + NW * FR;
    h16x8 *hfrag = wo_l + NW * FR;                  // B operands of the pos_proj.2 product
    h16x8 *kfrag = hfrag + FR;                      // ... of the K / V products (and of the Wq product at a group switch)
    h16x8 *ofrag = kfrag + FR;                      // ... of the Wo product
    // keys of the channel-wise max of the NEXT group's windows, [16 windows][C]; channel c of window w sits at
    // w C + ((c + 4 w) mod C): the 16 windows of a fragment read start in different banks
    int *qmax = reinterpret_cast<int *>(ofrag + FR);
#define CW_QSLOT(w_, c_) ((w_) * C + (((c_) + 4 * (w_)) & (C - 1)))
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    const int r = wv * RPW + lane / LPR, q = lane % LPR;  // row-wise view: row of a 16-row pass, channels [4 q, 4 q + 4)
    const int nw = *a.num_wins, n = a.num_voxels;
    if (nw <= 0) return;

    // ---- weights (requested first: they travel under the search below) ---------------------------------------------------
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
    }
    // Wq | Wo fragments of every wave -> LDS, lane-linear (2 NW FR x 16 bytes): all 16 requests of a thread in flight at once,
    // written to LDS after the search's first exchange (its latency covers theirs)
    constexpr int WCP = 2 * NW * FR / T;  // 16
    h16x8 wcp[WCP];
    {
        const h16x8 *s2 = packed + (size_t)3 * NW * FR + threadIdx.x;
#pragma unroll
        for (int u = 0; u < WCP; ++u) wcp[u] = s2[u * T];
    }
    // ---- this workgroup's CHUNK: consecutive windows [Wa, Wb) = consecutive rows [Ra, Rb) holding 1 / gridDim of the cost
    // (cost before row r = 8 r + 5 pair_win[r]: 16 rows cost ~5000 clocks, the per-group steps ~2500 per 16 windows; windows are
    // numbered in row order).  4096 probes -> the first probe at or above c total / gridDim -> the next window START at or
    // after it.  Every workgroup evaluates both of its ends with the same function: the chunks tile the level.
    int Wa, Wb, Ra, Rb;
    {
        constexpr int NPB = 8, NPROBE = NPB * T;
        int *ps = reinterpret_cast<int *>(hfrag);  // [probes] (16 KB: hfrag | kfrag), results behind the max tile's start
        static_assert(NPROBE * 4 <= 2 * FR * 16, "probes fit the two fragment sets");
        int *res = qmax;
#pragma unroll
        for (int k = 0; k < NPB; ++k) {
            const int idx = threadIdx.x + k * T, row = (int)((long long)idx * n / NPROBE);
            ps[idx] = 8 * row + 5 * max(a.pair_win[row], 0);
        }
        if (threadIdx.x < 2) res[threadIdx.x] = NPROBE;
#pragma unroll
        for (int u = 0; u < WCP; ++u) wq_l[threadIdx.x + u * T] = wcp[u];
        __syncthreads();
        const long long total = 8ll * n + 5ll * nw;
        int ends_w[2], ends_r[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const long long tgt = total * ((int)blockIdx.x + e) / (int)gridDim.x;
#pragma unroll
            for (int k = 0; k < NPB; ++k) {
                const int idx = threadIdx.x + k * T;
                if (ps[idx] >= tgt && (idx == 0 || ps[idx - 1] < tgt)) atomicMin(res + e, idx);
            }
        }
        __syncthreads();
        // the next window START at or after each end's probe row: the first listed row whose window differs from the row before
        // it.  Both ends' requests leave together (one round trip); 64 rows at a time
        int r0[2], prev[2], pwl[2];
        bool search[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int cc = (int)blockIdx.x + e, idx = res[e];
            search[e] = cc > 0 && cc < (int)gridDim.x && idx < NPROBE;
            ends_w[e] = cc <= 0 ? 0 : nw;
            ends_r[e] = cc <= 0 ? 0 : n;
            r0[e] = search[e] ? (int)((long long)idx * n / NPROBE) : 0;
            prev[e] = r0[e] > 0 ? a.pair_win[r0[e] - 1] : -1;
            pwl[e] = r0[e] + lane < n ? a.pair_win[r0[e] + lane] : -1;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (!search[e]) continue;
            int pv = prev[e], pw = pwl[e];
            for (int rr = r0[e];;) {
                const unsigned long long hit = __ballot(pw >= 0 && pw != pv);
                if (hit) {
                    const int l = __ffsll((long long)hit) - 1;
                    ends_w[e] = __builtin_amdgcn_readlane(pw, l);
                    ends_r[e] = rr + l;
                    break;
                }
                // (64 rows of one window / of no window: carry the last listed window on)
                const unsigned long long any = __ballot(pw >= 0);
                if (any) pv = __builtin_amdgcn_readlane(pw, 63 - __clzll((long long)any));
                rr += MSSVT_WAVE;
                if (rr >= n) break;
                pw = rr + lane < n ? a.pair_win[rr + lane] : -1;
            }
        }
        Wa = ends_w[0]; Wb = ends_w[1]; Ra = ends_r[0]; Rb = ends_r[1];
    }
    if (Wa >= Wb || Ra >= Rb) return;  // (wave-uniform, the same in every wave)

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

    // ---- the channel-wise max of a GROUP of 16 windows (the query tokens, ref :370), one group ahead of the rows ------------
    // One pass = 32 rows from `mp` in the row-wise view (coalesced) -> integer atomics on the keys.  Every wave also looks at
    // the windows of the 32 rows itself: the pass that meets the first row of the next group ends the group (mdone) and
    // leaves mp there -- wave-uniform without an exchange.
    float4 mxa, mxb;
    int mpa, mpb, pwp;
    int mp = Ra;
    bool mdone = false;
#define CW_MAX_LOAD()                                                                                         \
    {                                                                                                         \
        const int ra_ = mp + r, rc_ = mp + 16 + r, rp_ = mp + (lane & 31);                                    \
        pwp = rp_ < Rb ? a.pair_win[rp_] : 0x7FFFFFFF;                                                        \
        mxa = mxb = make_float4(0.f, 0.f, 0.f, 0.f);                                                          \
        mpa = mpb = -1;                                                                                       \
        if (ra_ < Rb) {                                                                                       \
            mxa = *reinterpret_cast<const float4 *>(a.xhat + (size_t)ra_ * C + 4 * q);                        \
            mpa = a.pair_win[ra_];                                                                            \
        }                                                                                                     \
        if (rc_ < Rb) {                                                                                       \
            mxb = *reinterpret_cast<const float4 *>(a.xhat + (size_t)rc_ * C + 4 * q);                        \
            mpb = a.pair_win[rc_];                                                                            \
        }                                                                                                     \
    }
#define CW_MAX_PUT(w0_, w1_)                                                                                  \
    {                                                                                                         \
        if (mpa >= (w0_) && mpa < (w1_)) {                                                                    \
            const int w_ = mpa - (w0_);                                                                       \
            atomicMax(qmax + CW_QSLOT(w_, 4 * q), cw_key(mxa.x)); atomicMax(qmax + CW_QSLOT(w_, 4 * q + 1), cw_key(mxa.y)); \
            atomicMax(qmax + CW_QSLOT(w_, 4 * q + 2), cw_key(mxa.z)); atomicMax(qmax + CW_QSLOT(w_, 4 * q + 3), cw_key(mxa.w)); \
        }                                                                                                     \
        if (mpb >= (w0_) && mpb < (w1_)) {                                                                    \
            const int w_ = mpb - (w0_);                                                                       \
            atomicMax(qmax + CW_QSLOT(w_, 4 * q), cw_key(mxb.x)); atomicMax(qmax + CW_QSLOT(w_, 4 * q + 1), cw_key(mxb.y)); \
            atomicMax(qmax + CW_QSLOT(w_, 4 * q + 2), cw_key(mxb.z)); atomicMax(qmax + CW_QSLOT(w_, 4 * q + 3), cw_key(mxb.w)); \
        }                                                                                                     \
        const unsigned long long past_ = __ballot(pwp >= (w1_)); /* (rows in no window: -1, never past) */    \
        if (past_) {                                                                                          \
            mp += __ffsll((long long)past_) - 1; /* lanes 0-31 hold the 32 rows in order */                   \
            mdone = true;                                                                                     \
        } else {                                                                                              \
            mp += 32;                                                                                         \
        }                                                                                                     \
    }
    // initial keys of a group's windows: 0 (= key(0.0f): the zero padding takes part, ref :370) unless the list is full;
    // cnt_: win_cnt of window w0_ + threadIdx / LPR, requested a group earlier
#define CW_CNT_LOAD(w0_) a.win_cnt[min((w0_) + (int)threadIdx.x / LPR, nw - 1)]
#define CW_MAX_INIT(w0_, cnt_)                                                                                \
    {                                                                                                         \
        const int w_ = threadIdx.x / LPR; /* T / LPR = 16 windows */                                          \
        const int init_ = (w0_) + w_ < Wb && (cnt_) >= a.ns ? cw_key(-INFINITY) : 0;                          \
        *reinterpret_cast<int4 *>(qmax + w_ * C + 4 * q) = make_int4(init_, init_, init_, init_);             \
    }
    int cnt_pre = CW_CNT_LOAD(Wa);
    __syncthreads();  // (the search's scratch is dead)
    CW_MAX_INIT(Wa, cnt_pre)
    cnt_pre = CW_CNT_LOAD(Wa + 16);
    CSTAMP(0)  // weights, chunk search

    // rows of the MFMA view, requested 16 rows ahead: window, next row's window, cell, this wave's 16 channels of xhat
#define CW_ROW_LOAD(rb_, pw_, pn_, vi_, xs_)                                                                  \
    {                                                                                                         \
        const int row_ = min((rb_) + la, Rb - 1);                                                             \
        pw_ = a.pair_win[row_];                                                                               \
        pn_ = a.pair_win[min(row_ + 1, n - 1)];                                                               \
        vi_ = reinterpret_cast<const int4 *>(a.indices)[row_];                                                \
        xs_ = *reinterpret_cast<const float4 *>(a.xhat + (size_t)row_ * C + ch);                              \
    }
    int pw_n, pn_n;
    int4 vi_n;
    float4 xs_n;
    CW_ROW_LOAD(Ra, pw_n, pn_n, vi_n, xs_n)
    int gq = -1;  // the newest group whose q' exists (qpB; qpA: the group before it)
    int jo = 0;   // the oldest group whose output rows are not written yet
    f32x4 qpA = f32x4{0.f, 0.f, 0.f, 0.f}, qpB = qpA;
    float cm = 0.f, cs = 0.f;  // carried (max, sum, sum p V) of the window open at the end of the previous 16 rows
    f32x2 co01 = f32x2{0.f, 0.f}, co23 = co01;
    int cseg = -0x7FFFFFFF;

    for (int rb = Ra; rb < Rb; rb += 16) {
        const int row = min(rb + la, Rb - 1);
        const bool rlive = rb + la < Rb;
        const int pw = pw_n, pnext = pn_n;
        const int4 vi = vi_n;
        const float4 xs = xs_n;
        CW_ROW_LOAD(rb + 16, pw_n, pn_n, vi_n, xs_n)
        const bool rvalid = rlive && pw >= Wa && pw < Wb;  // (rows of a chunk are its windows' or in no window)
        const bool wend = row + 1 >= Rb || pnext != pw;    // the window's last row
        const int grow = (pw - Wa) >> 4, widx = (pw - Wa) & 15;
        // ---- group switch: the rows reach a group whose queries do not exist yet --------------------------------------------
        while (__ballot(rvalid && grow > gq) != 0ull) {
            const int gw0 = Wa + 16 * (gq + 1), gw1 = min(gw0 + 16, Wb);
            if (!mdone) {  // the rest of the group's rows (its max normally completes under the rows of the group before it)
                __syncthreads();  // (the keys' initial values are in place)
                do {
                    CW_MAX_LOAD()
                    CW_MAX_PUT(gw0, gw1)
                } while (!mdone);
            }
            __syncthreads();  // the keys are final; the previous rows' K / V products have read their fragments
            {                 // q_tok, this wave's 16 channels of the 16 windows -> B fragments (in the K / V fragments' place)
                const int4 k0 = *reinterpret_cast<const int4 *>(qmax + CW_QSLOT(la, ch));
                h16x4 hi, lo;
                cw_split4(cw_unkey(k0.x), cw_unkey(k0.y), cw_unkey(k0.z), cw_unkey(k0.w), hi, lo);
                reinterpret_cast<h16x4 *>(kfrag + (ks * 2) * 64 + lane)[hs] = hi;
                reinterpret_cast<h16x4 *>(kfrag + (ks * 2 + 1) * 64 + lane)[hs] = lo;
            }
            __syncthreads();
            CW_MAX_INIT(gw1, cnt_pre)  // the max tile turns to the group after this one
            cnt_pre = CW_CNT_LOAD(gw1 + 16);
            mdone = gw1 >= Wb;  // (no group behind the chunk's last one)
            // q' = scale log2(e) (Wq q_tok + bq): head w of the 16 queries, lane (la = window, g): channels 16 w + 4 g + i
            {
                f32x4 m = f32x4{bq.x, bq.y, bq.z, bq.w}, l = f32x4{0.f, 0.f, 0.f, 0.f}, k = l;
                const h16x8 *wf = wq_l + (size_t)wv * FR + lane;
#pragma unroll
                for (int P = 0; P < NP; ++P) {
                    const h16x8 bh = kfrag[(P * 2) * 64 + lane], bl = kfrag[(P * 2 + 1) * 64 + lane];
                    const h16x8 ah = wf[(P * 2) * 64], al = wf[(P * 2 + 1) * 64];
                    MFMA_H(m, ah, bh);
                    MFMA_H(l, ah, bl);
                    MFMA_H(k, al, bh);
                }
                qpA = qpB;
#pragma unroll
                for (int i = 0; i < 4; ++i) qpB[i] = __builtin_fmaf(l[i] + k[i], CW_INV, m[i]) * a.qscale;
#ifdef CW_DEBUG
                if (gw0 + la < gw1 && g_cw_dbg[1])
                    for (int i = 0; i < 4; ++i) g_cw_dbg[1][(size_t)(gw0 + la) * C + ch + i] = qpB[i];
#endif
            }
            ++gq;
#ifdef CW_STAMPS
            cs_acc[14] += 1;
#endif
        }
        CSTAMP(2)  // group switch (max rest, Q product)
        const bool mrun = !mdone;  // this iteration carries a pass of the next group's max (wave-uniform)
        if (mrun) CW_MAX_LOAD()
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
        f32x2 o01, o23;
        float m_, s_;
        {
            // q' of the row's window: the rows of 16 consecutive rows belong to the newest group or to the one before it.
            // Requested BEFORE the products, and with them every branch of this block: between the last MFMA and the first
            // read of an accumulator there must be NO control flow -- the compiler pads that distance with s_nop inside a basic
            // block only; behind a branch the sums were read a few clocks early, i.e. without their last term (1e-4 errors
            // that came and went with the instruction schedule)
            const int src = 4 * (16 * g + (rvalid ? widx : 0));
            const bool newest = grow == gq;
            float qr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) qr[i] = cw_float(__builtin_amdgcn_ds_bpermute(src, cw_bits(qpB[i])));
            if (__ballot(rvalid && !newest) != 0ull) {  // (wave-uniform: the rows straddle two groups)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float qa = cw_float(__builtin_amdgcn_ds_bpermute(src, cw_bits(qpA[i])));
                    qr[i] = newest ? qr[i] : qa;
                }
            }
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
            float sc = 0.f;
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc = __builtin_fmaf(__builtin_fmaf(kl[i] + kk[i], CW_INV, km[i]), qr[i], sc);
                o[i] = __builtin_fmaf(vl[i] + vk[i], CW_INV, vm[i]);
            }
            sc += lane_xor16(sc);
            sc += lane_xor32(sc);
            m_ = sc;
            s_ = 1.0f;
            o01 = f32x2{o[0], o[1]};
            o23 = f32x2{o[2], o[3]};
        }
        CSTAMP(8)  // K, V, scores
#ifdef CW_DEBUG
        if (rvalid && g == 0 && g_cw_dbg[2]) g_cw_dbg[2][(size_t)row * NW + wv] = m_;
#endif
        const int seg = rvalid ? pw : -1 - la;  // rows outside the chunk's windows: segments of their own, never published
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
        {
            // LEFT FOLD over the row lanes: ((x0 + x1) + x2) + ... in row order, whatever the position of the window in these 16
            // rows and wherever a window is cut by their end (the carried piece is the fold so far) -- the association, hence
            // every bit of the result, is independent of how the level is cut into chunks and 16-row pieces: a scene's
            // output does not depend on the scenes it shares a batch with.  Step k: the k-th row of every
            // piece takes in the fold of the rows before it from the lane below (row_shr:1; every DPP with all lanes
            // active: a DPP source lane must be active).  max(rows of a window in the piece) - 1 steps: 3.6 on average at
            // 160k points against 2.3 for a scan tree, whose shape would depend on the cut (77.6 against 70.2 us).  Measured
            // and rejected: pieces that end on window ends + the scan tree (a window of <= 16 rows is then never cut and the
            // tree's shape follows from its length alone): 10.7 % more pieces, 81.2 us.
            const int pg1 = CW_DPP_I(seg, 0x111);
            const unsigned int starts = (unsigned int)__ballot((la == 0) | (pg1 != seg)) & 0xFFFFu;  // (the same in the four lane rows)
            const int pos = la - (31 - __clz((int)(starts & ((2u << la) - 1u))));  // row's place inside its piece
            const bool cont0 = la == 0 && cseg == seg;  // row lane 0 continues the window open at the end of the previous piece
            // The weights are taken WITHOUT a reference exponent, p = 2^score, and the fold is then plain additions of p and
            // p V -- 5 DPP moves + 5 fused multiply-adds per step instead of a two-exponential merge of (max, sum, sum p V)
            // states (timing-only ablation: the merge fold was 15 of the kernel's 78 us).  Same quotient sum p V / sum p, and
            // cut-independent like the merge: p depends on the row alone.  Allowed while every score of the piece lies in
            // [-100, 100] (2^+-100: the sums of <= 32 rows neither overflow nor vanish; scores behind a LayerNorm are two
            // orders of magnitude smaller) and no window comes in with a moved reference; else the piece -- a wave-uniform
            // decision -- takes the merge form (tests: scores 60 x apart).  (m, s, o) = (reference, sum p, sum p V) in both.
            // (every lane is tested, also rows that belong to no window: their p is multiplied by 0, which must not be 0 x inf)
            if (__ballot(!(__builtin_fabsf(m_) <= 100.0f) || (cont0 && cm != 0.0f)) == 0ull) {
                const float pw2 = __builtin_amdgcn_exp2f(m_);
                const float ref = 0.0f;
                const float c0 = cont0 ? 1.0f : 0.0f;
                s_ = __builtin_fmaf(cs, c0, pw2);
                o01 = __builtin_elementwise_fma(co01, f32x2{c0, c0}, o01 * f32x2{pw2, pw2});
                o23 = __builtin_elementwise_fma(co23, f32x2{c0, c0}, o23 * f32x2{pw2, pw2});
                m_ = ref;
                for (int k = 1;; ++k) {
                    const bool mk = pos == k;
                    if (__ballot(mk) == 0ull) break;  // (wave-uniform)
                    const float kf = mk ? 1.0f : 0.0f;
                    const float a0_ = o01[0], a1_ = o01[1], a2_ = o23[0], a3_ = o23[1];
                    s_ = __builtin_fmaf(CW_DPP_F(s_, 0x111), kf, s_);
                    o01 = f32x2{__builtin_fmaf(CW_DPP_F(a0_, 0x111), kf, a0_), __builtin_fmaf(CW_DPP_F(a1_, 0x111), kf, a1_)};
                    o23 = f32x2{__builtin_fmaf(CW_DPP_F(a2_, 0x111), kf, a2_), __builtin_fmaf(CW_DPP_F(a3_, 0x111), kf, a3_)};
                }
            } else {
                CW_COMBINE(cont0, cm, cs, co01, co23)  // the piece carried in from the previous 16 rows joins row lane 0
#define CW_STEP(ctrl_, mg_)                                                                             \
    {                                                                                                   \
        const float pm_ = CW_DPP_F(m_, ctrl_), ps_ = CW_DPP_F(s_, ctrl_);                               \
        const float a0_ = o01[0], a1_ = o01[1], a2_ = o23[0], a3_ = o23[1];                             \
        const f32x2 p01_ = f32x2{CW_DPP_F(a0_, ctrl_), CW_DPP_F(a1_, ctrl_)}, p23_ = f32x2{CW_DPP_F(a2_, ctrl_), CW_DPP_F(a3_, ctrl_)}; \
        CW_COMBINE(mg_, pm_, ps_, p01_, p23_)                                                           \
    }
                for (int k = 1;; ++k) {
                    const bool mk = pos == k;
                    if (__ballot(mk) == 0ull) break;  // (wave-uniform)
                    CW_STEP(0x111, mk)
                }
            }
#undef CW_STEP
#undef CW_COMBINE
        }
        // ---- windows that end in these rows: head w of their attention output -> B fragment of the Wo product; a group
        // whose last window ends here gets its output rows; ends of the group after it wait for that product
        {
            const int pend = rvalid && wend ? grow : -1;
            const float inv = __builtin_amdgcn_rcpf(s_);  // (1 ulp; the three-launch form divides)
            h16x4 hi, lo;
            cw_split4(o01[0] * inv, o01[1] * inv, o23[0] * inv, o23[1] * inv, hi, lo);
            for (;;) {
                const bool mine = pend == jo;
                if (mine) {
                    reinterpret_cast<h16x4 *>(ofrag + (ks * 2) * 64 + 16 * g + widx)[hs] = hi;
                    reinterpret_cast<h16x4 *>(ofrag + (ks * 2 + 1) * 64 + 16 * g + widx)[hs] = lo;
                }
                const int ow0 = Wa + 16 * jo, ow1 = min(ow0 + 16, Wb);
                if (__ballot(mine && pw == ow1 - 1) == 0ull) break;
                CSTAMP(9)
                __syncthreads();  // every head of the group's windows is in place
                CSTAMP(12)
                {  // out = Wo o + bo, this wave's 16 channels of the group's windows
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
                    // (formed before the predicated store: no branch between the products and the reads of their sums)
                    const float4 res = make_float4(__builtin_fmaf(l[0] + k[0], CW_INV, m[0]), __builtin_fmaf(l[1] + k[1], CW_INV, m[1]),
                                                   __builtin_fmaf(l[2] + k[2], CW_INV, m[2]), __builtin_fmaf(l[3] + k[3], CW_INV, m[3]));
                    __builtin_amdgcn_sched_barrier(0);
                    if (ow0 + la < ow1) *reinterpret_cast<float4 *>(a.out + (size_t)(ow0 + la) * C + ch) = res;
                }
                ++jo;
                CSTAMP(13)  // O product
                if (__ballot(pend == jo) == 0ull) break;
                __syncthreads();  // (the product has read the fragments)
            }
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
        CSTAMP(9)  // scan, publish
        // the next group's max, 32 rows per 16 rows of the stream (requested at the top of the iteration)
        if (mrun) {
            const int gw0 = Wa + 16 * (gq + 1), gw1 = min(gw0 + 16, Wb);
            CW_MAX_PUT(gw0, gw1)
        }
        CSTAMP(10)  // atomics
#ifdef CW_STAMPS
        cs_acc[15] += 1;
#endif
    }
#ifdef CW_STAMPS
    if (lane == 0 && blockIdx.x % 32 == 0) {
        for (int k = 0; k < 16; ++k) g_cw_stamps[((blockIdx.x / 32) * 8 + wv) * 16 + k] = cs_acc[k];
    }
#endif
#undef CW_QSLOT
#undef CW_MAX_LOAD
#undef CW_MAX_PUT
#undef CW_MAX_INIT
#undef CW_CNT_LOAD
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
                                 const int *num_wins_dev, int win_capacity, const int *indices, const int *win_cnt,
                                 const int *pair_win, const float *host_voxel_size3,
                                 const float *host_range_min3, const float *host_win_size3, const float *xhat, const float *Wpos1,
                                 const float *bpos1, const float *bpos2, const float *bq, const float *bkv, const float *bo,
                                 const void *packed, float *out, void *stream) {
    if (!num_wins_dev || !indices || !win_cnt || !pair_win || !host_voxel_size3 ||
        !host_range_min3 || !host_win_size3 || !xhat || !Wpos1 || !bpos1 || !bpos2 || !bq || !bkv || !bo || !packed || !out ||
        max_num_win1 <= 0 || num_voxels < 0 || win_capacity <= 0 || z_ws <= 0)
        return MSSVT_E_BADARG;
    if (C != 128 || head_dim != 16 || max_num_win1 > 32 || z_ws > max_num_win1) return MSSVT_E_TOOLARGE;
    if (num_voxels == 0) return MSSVT_OK;
    CwArgs a;
    a.ns = max_num_win1; a.num_voxels = num_voxels; a.z_magic = 65536 / z_ws + 1; a.qscale = scale * CW_LOG2E;
    a.num_wins = num_wins_dev; a.indices = indices;
    a.win_cnt = win_cnt; a.pair_win = pair_win;
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
