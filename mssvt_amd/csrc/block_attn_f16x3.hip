// block_attn_f16x3.hip -- window attention of one MsSVT Block in ONE launch with fp32-accurate matrix products
// (ref: pcdet/models/model_utils/mssvt_utils.py:112-150, pcdet/models/backbones_3d/mssvt_backbone.py:260-295).
//
// The single-launch form of block_attn_bf16.hip (one wavefront per (window, head group), keys projected in the kernel,
// nothing handed over through HBM -- see that file for the operand layouts and the software pipeline, which are
// unchanged) with every MFMA operand split exactly into two fp16 halves instead of rounded to bf16:
//      v = hi + 2^-11 lo,   hi = fp16(v),  lo = fp16((v - hi) 2^11)         (22+ mantissa bits; lo scaled out of the subnormals)
//      sum a b = sum a_hi b_hi + 2^-11 (sum a_hi b_lo + sum a_lo b_hi)       three v_mfma_f32_16x16x32_f16, fp32 accumulate
// i.e. the error of the fp32 matrix instruction (measured against float64) at 3/16 of its cycles.
// That rate is what the fp32 path lacked: block_attn.hip re-associates the attention so that keys are never projected
// and pays for it with three launches and a 1 KiB-per-query hand-off (165 of its 254 MB per Block).  Accumulation,
// softmax, biases, the positional MLP and every output stay fp32.  The caller guarantees the fp16 range of tokens and
// projections from the parameters (mssvt_amd/fused.py), else it runs the fp32 kernels.
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
struct hx8 {  // a split operand fragment
    h16x8 hi, lo;
};
#define HX_SCALE 2048.0f
#define HX_INV (1.0f / 2048.0f)

#define HA_WAVES 4
#define HA_MAX_GROUPS 4

struct AttnHxArgs {
    int C, c0, heads, hd;
    float scale;
    int nq, K;
    const float *xhat;
    const int *num_wins, *perm, *q_off, *nq_valid, *num_rows;
    const float4 *qrow_meta;
    const int2 *qrow_src;
    const float4 *kmeta, *wcentre;
    const float *Wq, *bq, *Wkv, *bkv, *Wo, *bo, *Wp, *bp;
    float *attn;
    int row_capacity;
};
struct AttnHxPack {
    AttnHxArgs g[HA_MAX_GROUPS];
};

__device__ __forceinline__ hx8 pack8(const f32x4 a, const f32x4 b) {
    hx8 r;
    const fp16x2 h0 = __builtin_amdgcn_cvt_pkrtz(a[0], a[1]), h1 = __builtin_amdgcn_cvt_pkrtz(a[2], a[3]);
    const fp16x2 h2 = __builtin_amdgcn_cvt_pkrtz(b[0], b[1]), h3 = __builtin_amdgcn_cvt_pkrtz(b[2], b[3]);
    const fp16x2 l0 = __builtin_amdgcn_cvt_pkrtz((a[0] - (float)h0[0]) * HX_SCALE, (a[1] - (float)h0[1]) * HX_SCALE);
    const fp16x2 l1 = __builtin_amdgcn_cvt_pkrtz((a[2] - (float)h1[0]) * HX_SCALE, (a[3] - (float)h1[1]) * HX_SCALE);
    const fp16x2 l2 = __builtin_amdgcn_cvt_pkrtz((b[0] - (float)h2[0]) * HX_SCALE, (b[1] - (float)h2[1]) * HX_SCALE);
    const fp16x2 l3 = __builtin_amdgcn_cvt_pkrtz((b[2] - (float)h3[0]) * HX_SCALE, (b[3] - (float)h3[1]) * HX_SCALE);
    r.hi = h16x8{(_Float16)h0[0], (_Float16)h0[1], (_Float16)h1[0], (_Float16)h1[1], (_Float16)h2[0], (_Float16)h2[1],
                 (_Float16)h3[0], (_Float16)h3[1]};
    r.lo = h16x8{(_Float16)l0[0], (_Float16)l0[1], (_Float16)l1[0], (_Float16)l1[1], (_Float16)l2[0], (_Float16)l2[1],
                 (_Float16)l3[0], (_Float16)l3[1]};
    return r;
}
// acc += a_hi b_hi ; lacc += a_hi b_lo + a_lo b_hi ; the product sum is acc + 2^-11 lacc
#define MFMA_HX(acc, lacc, av, bv)                                                          \
    {                                                                                       \
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16((av).hi, (bv).hi, acc, 0, 0, 0);       \
        lacc = __builtin_amdgcn_mfma_f32_16x16x32_f16((av).hi, (bv).lo, lacc, 0, 0, 0);     \
        lacc = __builtin_amdgcn_mfma_f32_16x16x32_f16((av).lo, (bv).hi, lacc, 0, 0, 0);     \
    }
__device__ __forceinline__ f32x4 hx_fin(const f32x4 m, const f32x4 l) {
    return f32x4{__builtin_fmaf(l[0], HX_INV, m[0]), __builtin_fmaf(l[1], HX_INV, m[1]), __builtin_fmaf(l[2], HX_INV, m[2]),
                 __builtin_fmaf(l[3], HX_INV, m[3])};
}
#define MFMA_F4(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av), (bv), acc, 0, 0, 0)

#ifdef MSSVT_STAMPS  // developer instrumentation: cycles per phase, summed per wave (first 64 workgroups, group 0)
__device__ unsigned long long g_attn_hx_stamps[64 * HA_WAVES * 8];
extern "C" int mssvt_debug_read_attn_hx_stamps(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_hx_stamps), sizeof(g_attn_hx_stamps));
}
#define BSTAMP(k_) { const unsigned long long now_ = __builtin_readcyclecounter(); ph[k_] += now_ - tlast; tlast = now_; }
#else
#define BSTAMP(k_)
#endif

template <int CG, int HD, int KT>
__global__ void __launch_bounds__(HA_WAVES *MSSVT_WAVE, KT >= 4 ? 1 : 2) k_attn_f16x3(AttnHxPack pack) {
    const AttnHxArgs &a = pack.g[blockIdx.y];
    constexpr int CGP = (CG + 15) / 16 * 16, NT = CGP / 16, NS = (NT + 1) / 2, KS = (KT + 1) / 2;
    constexpr int NH = CG / HD, HP = NH <= 1 ? 1 : (NH <= 2 ? 2 : (NH <= 4 ? 4 : 8)), QPP = 16 / HP;
    constexpr f32x4 Z4 = {0.f, 0.f, 0.f, 0.f};
    extern __shared__ float4 lds4[];
    h16x8 *Wfh = reinterpret_cast<h16x8 *>(lds4);                        // [4][NT][NS][64] fragments, hi halves
    h16x8 *Wfl = Wfh + 4 * NT * NS * 64;                                 // ... lo halves
    float *bias_l = reinterpret_cast<float *>(Wfl + 4 * NT * NS * 64);   // [3][CGP]: bq, bv, bo
#define WF(idx_) hx8{Wfh[idx_], Wfl[idx_]}
    float4 *wpc_l = reinterpret_cast<float4 *>(bias_l + 3 * CGP);        // [CGP]: (wp3, wp4, wp5, bp) of the channel
    // ---- stage the four matrices as bf16 fragments (mat 0 Wq, 1 Wk, 2 Wv, 3 Wo) -------------------------
    {
        constexpr int NF = 4 * NT * NS * 64, PER = (NF + HA_WAVES * MSSVT_WAVE - 1) / (HA_WAVES * MSSVT_WAVE);
        float4 lo[PER], hi[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int f = threadIdx.x + u * HA_WAVES * MSSVT_WAVE;
            const int fl = f & 63, fs = (f >> 6) % NS, fn = ((f >> 6) / NS) % NT, mat = (f >> 6) / (NS * NT);
            const int row = 16 * fn + (fl & 15), fg = fl >> 4;
            const float *W = mat == 0 ? a.Wq : (mat == 1 ? a.Wkv : (mat == 2 ? a.Wkv + (size_t)CG * CG : a.Wo));
            const int cl = 16 * (2 * fs) + 4 * fg, ch = 16 * (2 * fs + 1) + 4 * fg;
            const bool ok = f < NF && mat < 4 && row < CG;
            lo[u] = ok && cl < CG ? *reinterpret_cast<const float4 *>(W + (size_t)row * CG + cl)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            hi[u] = ok && ch < CG ? *reinterpret_cast<const float4 *>(W + (size_t)row * CG + ch)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int f = threadIdx.x + u * HA_WAVES * MSSVT_WAVE;
            if (f < NF) {
                const hx8 w = pack8(f32x4{lo[u].x, lo[u].y, lo[u].z, lo[u].w}, f32x4{hi[u].x, hi[u].y, hi[u].z, hi[u].w});
                Wfh[f] = w.hi;
                Wfl[f] = w.lo;
            }
        }
        for (int e = threadIdx.x; e < 3 * CGP; e += blockDim.x) {
            const int which = e / CGP, c = e % CGP;
            bias_l[e] = c < CG ? (which == 0 ? a.bq[c] : (which == 1 ? a.bkv[CG + c] : a.bo[c])) : 0.f;
        }
        for (int c = threadIdx.x; c < CGP; c += blockDim.x) {
            const float *wp = a.Wp + (size_t)(a.c0 + (c < CG ? c : 0)) * 6;
            wpc_l[c] = c < CG ? make_float4(wp[3], wp[4], wp[5], a.bp[a.c0 + c]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = threadIdx.x / MSSVT_WAVE;
    // positional MLP operand of this lane (channel 16 u + la, input g): lanes g < 3 hold the weight of rel. coordinate g;
    // lanes g == 3 the window part (bp + wp3..5 . centre), rebuilt per window from LDS
    float wrel[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int c = 16 * u + la;
        const bool in = (CGP == CG || c < CG) && g < 3;
        wrel[u] = in ? a.Wp[(size_t)(a.c0 + c) * 6 + g] : 0.f;
    }
    const int hh = la % HP, ql = la / HP;  // this column's head and query of the pass
    // per-lane multipliers that keep a column's own head: rows 16 n + 4 g .. of tile n belong to head (16 n + 4 g) / HD.
    // The softmax runs on base-2 exponentials: log2(e) rides on the query scale.
    float qmul[NT], omul[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const bool mine = (16 * n + 4 * g) / HD == hh;
        qmul[n] = mine ? a.scale * 1.4426950408889634f : 0.f;
        omul[n] = mine ? 1.0f : 0.f;
    }
    // gathers go through buffer descriptors: 32-bit lane offsets, no 64-bit address arithmetic per load (xhat rows:
    // row * C * 4 + this lane's 16-byte piece; the immediate offset walks the channel tiles)
    const __amdgpu_buffer_rsrc_t xr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.xhat), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t km_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(a.kmeta), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t qm_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(a.qrow_meta), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t qs_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int2 *>(a.qrow_src), 0, -1, 0x00020000);
    const unsigned row_bytes = (unsigned)a.C * 4u, lane_off = ((unsigned)a.c0 + 4u * g) * 4u;
#define BF_ROW4(off_, S_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr_rs, (off_) + 64u * (S_), 0, 0))
    const int n_act = __builtin_amdgcn_readfirstlane(*a.num_wins);
    const int wstep = gridDim.x * HA_WAVES;
    const int K = a.K;
    int wi = __builtin_amdgcn_readfirstlane(blockIdx.x * HA_WAVES + wv);  // wave-uniform: the metadata below is scalar
    if (wi >= n_act) return;
#ifdef MSSVT_STAMPS
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    // Software pipeline over this wave's windows, every load of the loop body UNCONDITIONAL (indices clamped to the
    // last window): a guarded load makes the wait counts path dependent and the compiler falls back to vmcnt(0),
    // which drains the prefetches.  vmcnt retires in order, so a wait for a load also waits for every load issued
    // before it -- hence the order of the stages inside a step:
    //   stage P  window id (scalar load)                         three steps ahead
    //   stage M  its metadata: centre / counts (scalar), key slots (vector)   two steps ahead
    //   stage R  raw key rows + the first pass's query metadata   one step ahead
    int w_p;
    float4 wc_m, km_m[KT];
    int nqv_m, qbase_m;
#define BF_LOAD_META()                                                                     \
    {                                                                                      \
        wc_m = a.wcentre[w_p];                                                             \
        nqv_m = a.nq_valid[w_p];                                                           \
        qbase_m = a.q_off[w_p];                                                            \
        _Pragma("unroll") for (int t = 0; t < KT; ++t)                                     \
            km_m[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(    \
                km_rs, ((unsigned)w_p * (unsigned)K + (unsigned)min(16 * t + la, K - 1)) * 16u, 0, 0)); \
    }
    float4 wc_r, qm_r;
    int2 qs_r;
    int nqv_r, qbase_r;
    float rel_r[KT];
    unsigned vmask_r, used_r;
    f32x4 T1n[KT][NT];
#define BF_ISSUE_ROWS()                                                                    \
    {                                                                                      \
        wc_r = wc_m; qbase_r = qbase_m;                                                    \
        nqv_r = qbase_m + nqv_m <= a.row_capacity ? nqv_m : 0;                             \
        vmask_r = 0; used_r = 0;                                                           \
        _Pragma("unroll") for (int t = 0; t < KT; ++t) {                                   \
            const int r_ = __builtin_bit_cast(int, km_m[t].w);                             \
            const bool ok_ = 16 * t + la < K && r_ >= 0;                                   \
            const unsigned long long bal_ = __ballot(ok_);                                 \
            vmask_r |= (unsigned)((bal_ >> (4 * g)) & 15ull) << (4 * t);                   \
            used_r |= (t == 0 || (bal_ & 0xFFFFull) != 0ull) ? 1u << t : 0u;               \
            rel_r[t] = lane_pick4(g, km_m[t].x, km_m[t].y, km_m[t].z, 1.0f); \
            /* rows of empty slots / unused tiles read row 0 (never used; a guarded load would cost the pipeline) */ \
            const unsigned ro_ = (unsigned)__umul24((unsigned)(ok_ ? r_ : 0), row_bytes) + lane_off; \
            _Pragma("unroll") for (int S = 0; S < NT; ++S)                                 \
                T1n[t][S] = (CGP == CG || 16 * S + 4 * g < CG) ? BF_ROW4(ro_, S) : Z4;     \
        }                                                                                  \
        const unsigned qr_ = (unsigned)min(qbase_r + min(ql, max(nqv_r, 1) - 1), a.row_capacity - 1); \
        qm_r = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(qm_rs, qr_ * 16u, 0, 0)); \
        qs_r = __builtin_bit_cast(int2, __builtin_amdgcn_raw_buffer_load_b64(qs_rs, qr_ * 8u, 0, 0)); \
    }
    const int w_last = n_act - 1;
    w_p = a.perm[wi];
    BF_LOAD_META()
    w_p = a.perm[min(wi + wstep, w_last)];
    BF_ISSUE_ROWS()
    BF_LOAD_META()
    w_p = a.perm[min(wi + 2 * wstep, w_last)];
    BSTAMP(0)
    for (; wi < n_act; wi += wstep) {
#ifdef MSSVT_STAMPS
        ph[7] += 1;
#endif
        const float4 wc = wc_r;
        const int nqv = nqv_r, qbase = qbase_r;
        const unsigned vmask = vmask_r, used = used_r;
        float4 qm = qm_r;
        int2 qs = qs_r;
        // key tokens: + relu(positional MLP), rounded to bf16 operands (A of Vp = T Wv^T, B of Kp^T = Wk T^T)
        float wu[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float4 pc = wpc_l[16 * u + la];
            wu[u] = g == 3 ? ((pc.w + pc.x * wc.x) + pc.y * wc.y) + pc.z * wc.z : wrel[u];
        }
        hx8 Tb[KT][NS];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            if (!(used >> t & 1)) continue;
            f32x4 tk[2 * NS];
#pragma unroll
            for (int u = 0; u < 2 * NS; ++u) {
                tk[u] = Z4;
                if (u < NT) {
                    f32x4 p1 = Z4;
                    MFMA_F4(p1, wu[u], rel_r[t]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) tk[u][i] = T1n[t][u][i] + fmaxf(p1[i], 0.0f);
                }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) Tb[t][s] = pack8(tk[2 * s], tk[2 * s + 1]);
        }
        BSTAMP(1)
        // the first pass's query rows: requested only now -- the wait for the key rows above must not cover them
        f32x4 xq[NT];
        {
            const unsigned ro = (unsigned)__umul24((unsigned)__builtin_bit_cast(int, qm.w), row_bytes) + lane_off;
#pragma unroll
            for (int S = 0; S < NT; ++S) xq[S] = (CGP == CG || 16 * S + 4 * g < CG) ? BF_ROW4(ro, S) : Z4;
        }
        // ---- next window: rows in flight under this window's MFMAs, metadata one further ahead (past the end of
        // the work list the last window is fetched again: unconditional loads keep the wait counts exact) --------
        BF_ISSUE_ROWS()
        BF_LOAD_META()  // id loaded one step ago
        w_p = a.perm[min(wi + 3 * wstep, w_last)];
        BSTAMP(2)
        // ---- key / value projections ---------------------------------------------------------------------
        // tile n of the outputs at a time, for every used key tile: the weight fragments of tile n + 1 are read
        // from LDS while tile n multiplies (an unpipelined ds_read -> MFMA pair exposes ~100 cycles per group)
        hx8 Kb[KT][NS];  // A operand of S = Kp Q'm^T: lane (key la, g), k-slots = Kp channels
        hx8 Vb[NT][KS];  // A operand of O^T = Vp^T P: lane (channel la of tile n, g), k-slots = keys
        {
            hx8 wk[NS], wv[NS], wkn[NS], wvn[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                wk[s] = WF(((1 * NT + 0) * NS + s) * 64 + lane);
                wv[s] = WF(((2 * NT + 0) * NS + s) * 64 + lane);
            }
            f32x4 kacc[KT][2], kaccl[KT][2];
#pragma unroll
            for (int n = 0; n < 2 * NS; ++n) {
                if (n < NT) {
                    if (n + 1 < NT) {
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            wkn[s] = WF(((1 * NT + n + 1) * NS + s) * 64 + lane);
                            wvn[s] = WF(((2 * NT + n + 1) * NS + s) * 64 + lane);
                        }
                    }
                    f32x4 vacc[2 * KS], vaccl[2 * KS];
#pragma unroll
                    for (int t = 0; t < 2 * KS; ++t) vacc[t] = vaccl[t] = Z4;
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        kacc[t][n & 1] = kaccl[t][n & 1] = Z4;
                        if (!(used >> t & 1)) continue;
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            MFMA_HX(kacc[t][n & 1], kaccl[t][n & 1], wk[s], Tb[t][s]);
                            MFMA_HX(vacc[t], vaccl[t], Tb[t][s], wv[s]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        Vb[n][s] = pack8(hx_fin(vacc[2 * s], vaccl[2 * s]), hx_fin(vacc[2 * s + 1], vaccl[2 * s + 1]));
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        wk[s] = wkn[s];
                        wv[s] = wvn[s];
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < KT; ++t) kacc[t][n & 1] = kaccl[t][n & 1] = Z4;
                }
                if (n & 1) {
#pragma unroll
                    for (int t = 0; t < KT; ++t)
                        Kb[t][n >> 1] = pack8(hx_fin(kacc[t][0], kaccl[t][0]), hx_fin(kacc[t][1], kaccl[t][1]));
                }
            }
        }
        // ---- queries, QPP per pass: column la = query * HP + head ----------------------------------------
        BSTAMP(3)
        float4 qm_n = qm;
        int2 qs_n = qs;
        for (int q0 = 0; q0 < nqv; q0 += QPP) {
            // the next pass's query metadata is requested now, its rows once this pass has consumed xq: both round
            // trips of a further pass run under this pass's products
            const bool more = q0 + QPP < nqv;
            if (q0 > 0) {
                qm = qm_n;
                qs = qs_n;
            }
            if (more) {
                const unsigned qr = (unsigned)(qbase + min(q0 + QPP + ql, nqv - 1));
                qm_n = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(qm_rs, qr * 16u, 0, 0));
                qs_n = __builtin_bit_cast(int2, __builtin_amdgcn_raw_buffer_load_b64(qs_rs, qr * 8u, 0, 0));
            }
            hx8 wq[NS], wqn[NS];  // Wq fragments of output tile 0: in flight under the positional MLP
#pragma unroll
            for (int s = 0; s < NS; ++s) wq[s] = WF(((0 * NT + 0) * NS + s) * 64 + lane);
            // query tokens: + relu(positional MLP) -> B operand of Q'^T = Wq Xq^T
            const float qrel = lane_pick4(g, qm.x, qm.y, qm.z, 1.0f);
            hx8 Xb[NS];
            {
                f32x4 tk[2 * NS];
#pragma unroll
                for (int u = 0; u < 2 * NS; ++u) {
                    tk[u] = Z4;
                    if (u < NT) {
                        f32x4 p1 = Z4;
                        MFMA_F4(p1, wu[u], qrel);
#pragma unroll
                        for (int i = 0; i < 4; ++i) tk[u][i] = xq[u][i] + fmaxf(p1[i], 0.0f);
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) Xb[s] = pack8(tk[2 * s], tk[2 * s + 1]);
            }
            if (more) {  // xq is dead: the next pass's rows travel from here on
                const unsigned ro = (unsigned)__umul24((unsigned)__builtin_bit_cast(int, qm_n.w), row_bytes) + lane_off;
#pragma unroll
                for (int S = 0; S < NT; ++S) xq[S] = (CGP == CG || 16 * S + 4 * g < CG) ? BF_ROW4(ro, S) : Z4;
            }
            BSTAMP(4)
            // Q'^T[o][col] = sum_c Wq[o][c] xq[col][c] + bq[o]; scaled; rows outside the column's head -> 0
            hx8 Qb[NS];
            {
                f32x4 qa[2], qal[2];
#pragma unroll
                for (int n = 0; n < 2 * NS; ++n) {
                    qa[n & 1] = qal[n & 1] = Z4;
                    if (n < NT) {
                        if (n + 1 < NT) {
#pragma unroll
                            for (int s = 0; s < NS; ++s) wqn[s] = WF(((0 * NT + n + 1) * NS + s) * 64 + lane);
                        }
                        const float4 b = *reinterpret_cast<const float4 *>(bias_l + 16 * n + 4 * g);
                        qa[n & 1] = f32x4{b.x, b.y, b.z, b.w};
#pragma unroll
                        for (int s = 0; s < NS; ++s) MFMA_HX(qa[n & 1], qal[n & 1], wq[s], Xb[s]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) qa[n & 1][i] = __builtin_fmaf(qal[n & 1][i], HX_INV, qa[n & 1][i]) * qmul[n];
#pragma unroll
                        for (int s = 0; s < NS; ++s) wq[s] = wqn[s];
                    }
                    if (n & 1) Qb[n >> 1] = pack8(qa[0], qa[1]);
                }
            }
            // scores S[key][col] = sum_o Kp[key][o] Q'm[col][o]
            f32x4 sc[2 * KS];
            {
                f32x4 scl[2 * KS];
#pragma unroll
                for (int t = 0; t < 2 * KS; ++t) sc[t] = scl[t] = Z4;
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    if (!(used >> t & 1)) continue;
#pragma unroll
                    for (int s = 0; s < NS; ++s) MFMA_HX(sc[t], scl[t], Kb[t][s], Qb[s]);
                }
#pragma unroll
                for (int t = 0; t < 2 * KS; ++t) sc[t] = hx_fin(sc[t], scl[t]);
            }
            // softmax over the unmasked keys: lane (col, g) holds keys 16 t + 4 g + i
            // masked slots and unused tiles score -inf: exp2(-inf) = 0, no branch (slot 0 is never masked)
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2 * KS; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[t][i] = (t < KT && (vmask >> (4 * t + i) & 1)) ? sc[t][i] : -INFINITY;
                    mx = fmaxf(mx, sc[t][i]);
                }
            mx = fmaxf(mx, lane_xor16(mx));
            mx = fmaxf(mx, lane_xor32(mx));
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < 2 * KS; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = __builtin_amdgcn_exp2f(sc[t][i] - mx);
                    sc[t][i] = e;
                    sum += e;
                }
            sum += lane_xor16(sum);
            sum += lane_xor32(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);  // slot 0 of a list is never masked: sum >= 1
            hx8 Pb[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                f32x4 p0, p1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    p0[i] = sc[2 * s][i] * inv;
                    p1[i] = sc[2 * s + 1][i] * inv;
                }
                Pb[s] = pack8(p0, p1);
            }
            hx8 wo[NS], won[NS];  // Wo fragments of output tile 0: in flight under the PV product
#pragma unroll
            for (int s = 0; s < NS; ++s) wo[s] = WF(((3 * NT + 0) * NS + s) * 64 + lane);
            // O^T[o][col] = sum_key Vp[key][o] P[key][col] + bv[o]; rows outside the column's head -> 0
            hx8 Ob[NS];
            {
                f32x4 oa[2 * NS];
#pragma unroll
                for (int n = 0; n < 2 * NS; ++n) {
                    oa[n] = Z4;
                    if (n < NT) {
                        const float4 b = *reinterpret_cast<const float4 *>(bias_l + CGP + 16 * n + 4 * g);
                        oa[n] = f32x4{b.x, b.y, b.z, b.w};
                        f32x4 oal = Z4;
#pragma unroll
                        for (int s = 0; s < KS; ++s) MFMA_HX(oa[n], oal, Vb[n][s], Pb[s]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) oa[n][i] = __builtin_fmaf(oal[i], HX_INV, oa[n][i]) * omul[n];
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) Ob[s] = pack8(oa[2 * s], oa[2 * s + 1]);
            }
            // out^T[p][col] = sum_{o in head(col)} Wo[p][o] O[col][o]; summed over the HP columns of the query
            const bool q_ok = q0 + ql < nqv;
            float *dst = a.attn + (size_t)qs.y * a.C + a.c0;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if (n + 1 < NT) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) won[s] = WF(((3 * NT + n + 1) * NS + s) * 64 + lane);
                }
                f32x4 acc = Z4, accl = Z4;
#pragma unroll
                for (int s = 0; s < NS; ++s) MFMA_HX(acc, accl, wo[s], Ob[s]);
                __builtin_amdgcn_sched_barrier(0);
                acc = hx_fin(acc, accl);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[i];
                    if (HP >= 2) v += DPP_MOV(v, 0xB1);   // lane ^ 1
                    if (HP >= 4) v += DPP_MOV(v, 0x4E);   // lane ^ 2
                    if (HP >= 8) v += DPP_MOV(v, 0x141);  // the other quad of the 8-lane group
                    acc[i] = v;
                }
                if (q_ok && (n % HP) == hh && (CGP == CG || 16 * n + 4 * g < CG)) {
                    const float4 b = *reinterpret_cast<const float4 *>(bias_l + 2 * CGP + 16 * n + 4 * g);
                    *reinterpret_cast<float4 *>(dst + 16 * n + 4 * g) =
                        make_float4(acc[0] + b.x, acc[1] + b.y, acc[2] + b.z, acc[3] + b.w);
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) wo[s] = won[s];
            }
            BSTAMP(5)
        }
    }
#ifdef MSSVT_STAMPS
    if (lane == 0 && blockIdx.x < 64 && blockIdx.y == 0) {
        for (int k = 0; k < 8; ++k) g_attn_hx_stamps[(blockIdx.x * HA_WAVES + wv) * 8 + k] = ph[k];
    }
#endif
#undef BF_LOAD_META
#undef BF_ISSUE_ROWS
#undef BF_ROW4
#undef WF
}

template <int CG, int HD>
static int launch_attn_f16x3(const AttnHxPack &pack, int ng, hipStream_t stream) {
    constexpr int CGP = (CG + 15) / 16 * 16, NT = CGP / 16, NS = (NT + 1) / 2;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int K = pack.g[0].K;
    // persistent over the work order: 2 workgroups of 4 waves per CU (VGPR bound)
    const dim3 grid(cus * 2 / ng > 0 ? cus * 2 / ng : 1, ng);
    const size_t lds = (size_t)2 * 4 * NT * NS * 64 * 16 + (size_t)3 * CGP * 4 + (size_t)CGP * 16;
#define HX_LAUNCH(KT_)                                                                                       \
    {                                                                                                        \
        if (lds > 64 * 1024) {                                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_attn_f16x3<CG, HD, KT_>),    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
            if (e != hipSuccess) return (int)e;                                                              \
        }                                                                                                    \
        k_attn_f16x3<CG, HD, KT_><<<grid, HA_WAVES * MSSVT_WAVE, lds, stream>>>(pack);                       \
    }
    if (K <= 16) HX_LAUNCH(1)
    else if (K <= 32) HX_LAUNCH(2)
    else HX_LAUNCH(4)
#undef HX_LAUNCH
    return mssvt_launch_status();
}

static int dispatch_attn_f16x3(const AttnHxPack &pack, int ng, int Cg, int head_dim, hipStream_t st) {
#define MSSVT_ATTN_HX_CASE(cg, hd) \
    if (Cg == cg && head_dim == hd) return launch_attn_f16x3<cg, hd>(pack, ng, st);
    MSSVT_ATTN_HX_CASE(16, 8)
    MSSVT_ATTN_HX_CASE(16, 16)
    MSSVT_ATTN_HX_CASE(32, 8)
    MSSVT_ATTN_HX_CASE(32, 16)
    MSSVT_ATTN_HX_CASE(32, 32)
    MSSVT_ATTN_HX_CASE(48, 16)
    MSSVT_ATTN_HX_CASE(64, 8)
    MSSVT_ATTN_HX_CASE(64, 16)
    MSSVT_ATTN_HX_CASE(64, 32)
    return MSSVT_E_TOOLARGE;  // shape not instantiated: the caller uses the fp32 kernels
#undef MSSVT_ATTN_HX_CASE
}

extern "C" int mssvt_block_attention_f16x3(
    int C, int num_groups, const int *host_c0, const int *host_cg, const int *host_heads, int head_dim, float scale,
    int nq, int key_num_sample, const float *xhat, const int *num_active_dev, const int *perm, const int *q_off,
    const int *nq_valid, const int *num_rows_dev, int row_capacity, const float *qrow_meta, const int *qrow_src,
    const float *const *host_kmeta, const float *wcentre, const float *const *host_Wq, const float *const *host_bq,
    const float *const *host_Wkv, const float *const *host_bkv, const float *const *host_Wo,
    const float *const *host_bo, const float *Wpos, const float *bpos, float *attn, void *stream) {
    if (!host_c0 || !host_cg || !host_heads || !xhat || !num_active_dev || !perm || !q_off || !nq_valid ||
        !num_rows_dev || !qrow_meta || !qrow_src || !host_kmeta || !wcentre || !host_Wq || !host_bq || !host_Wkv ||
        !host_bkv || !host_Wo || !host_bo || !Wpos || !bpos || !attn || C <= 0 || num_groups <= 0 ||
        head_dim <= 0 || nq <= 0 || key_num_sample <= 0 || row_capacity <= 0)
        return MSSVT_E_BADARG;
    if (C & 3) return MSSVT_E_BADARG;
    if (key_num_sample > MSSVT_WAVE || (head_dim & 3)) return MSSVT_E_TOOLARGE;
    hipStream_t st = (hipStream_t)stream;
    bool same = num_groups <= HA_MAX_GROUPS;
    for (int g = 0; g < num_groups; ++g) same = same && host_cg[g] == host_cg[0];
    AttnHxPack pack;
    for (int g = 0; g < num_groups; ++g) {
        const int c0 = host_c0[g], Cg = host_cg[g], heads = host_heads[g];
        if (!host_kmeta[g] || !host_Wq[g] || !host_bq[g] || !host_Wkv[g] || !host_bkv[g] || !host_Wo[g] || !host_bo[g])
            return MSSVT_E_BADARG;
        if (Cg <= 0 || heads <= 0 || Cg != heads * head_dim || c0 < 0 || c0 + Cg > C || (c0 & 3)) return MSSVT_E_BADARG;
        if (Cg > MSSVT_WAVE || heads > 8) return MSSVT_E_TOOLARGE;
        AttnHxArgs a;
        a.C = C; a.c0 = c0; a.heads = heads; a.hd = head_dim; a.scale = scale;
        a.nq = nq; a.K = key_num_sample;
        a.xhat = xhat; a.num_wins = num_active_dev; a.perm = perm; a.q_off = q_off; a.nq_valid = nq_valid;
        a.num_rows = num_rows_dev;
        a.qrow_meta = reinterpret_cast<const float4 *>(qrow_meta);
        a.qrow_src = reinterpret_cast<const int2 *>(qrow_src);
        a.kmeta = reinterpret_cast<const float4 *>(host_kmeta[g]);
        a.wcentre = reinterpret_cast<const float4 *>(wcentre);
        a.Wq = host_Wq[g]; a.bq = host_bq[g]; a.Wkv = host_Wkv[g]; a.bkv = host_bkv[g];
        a.Wo = host_Wo[g]; a.bo = host_bo[g]; a.Wp = Wpos; a.bp = bpos;
        a.attn = attn;
        a.row_capacity = row_capacity;
        if (same) {
            pack.g[g] = a;
        } else {  // unequal group widths: one launch per group
            AttnHxPack one;
            for (int i = 0; i < HA_MAX_GROUPS; ++i) one.g[i] = a;
            const int rc = dispatch_attn_f16x3(one, 1, Cg, head_dim, st);
            if (rc) return rc;
        }
    }
    if (!same) return MSSVT_OK;
    for (int g = num_groups; g < HA_MAX_GROUPS; ++g) pack.g[g] = pack.g[0];
    return dispatch_attn_f16x3(pack, num_groups, host_cg[0], head_dim, st);
}
