// compress_fused.hip -- CompressBlock attention on the fp32 matrix cores, four launches, no host sync.
//
// Fast form of the reference's MixedScaleSparseTransformerCompressBlock.forward up to (not including)
// the FFN tail (ref: mssvt_backbone.py:351-383, MixedScaleAttention mssvt_utils.py:112-150) for the
// common configuration: ONE head group, window lists that do not overlap (every voxel is a key of
// exactly one window: pair row = voxel row, see k_window_plan_one).  Every count is read on the
// device; the caller synchronises once, after the whole block is enqueued, to learn the output size.
//
//   A  (k_cmp_query_keys, first workgroups) : rows = windows.  q_tok = channel-wise max over the window's (zero padded) key
//                    features (ref :370); q' = scale (Wq q_tok + bq)                       -> qp (nw, C)
//   B  (k_cmp_query_keys, the others) : rows = voxels.   h = relu(Wp1 [rel ; centre] + bp1)   (one K = 8 MFMA per tile)
//                    k_tok = xhat + relu(Wp2 h + bp2)                                      -> ktok (N, C)
//   C  k_cmp_kv    : rows = voxels.   K = Wk k_tok + bk  -> score[v][head] = q'[window(v)] . K[v]
//                                     V = Wv k_tok + bv                                    -> vp (N, C)
//   D  k_cmp_out   : rows = windows.  softmax over the window's voxels, o = sum_v p_v V[v];
//                    new = Wo o + bo                                                       -> (nw, C)
// All of them are row-tiled GEMMs in the style of block_attn.hip / ffn.hip: one wavefront = 16 rows,
// products computed transposed (A = weight rows from LDS, one ds_read_b128 per 4 k-steps; B = the
// activations, lane (row = l % 16, g = l / 16) -> channels 16 S + 4 g + j = 16-byte global accesses),
// weights resident in LDS for the whole launch, no barrier after staging.
// Masked list slots carry an additive -100 in the reference (relative weight <= e^-100): skipped.
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// waves per workgroup: ONE workgroup per CU (the 66 KiB weight matrix is staged once per CU; two 8-wave workgroups do
// not fit beside the per-wave window lists), as many waves as the registers allow
#define CFQ_NW 12  // k_cmp_query_keys (<= 168 VGPRs)
#define CFK_NW 12  // k_cmp_kv
#define CFO_NW 16  // k_cmp_out (<= 128 VGPRs)
#define MFMA4(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av), (bv), acc, 0, 0, 0)

struct CmpArgs {
    int C, ns, num_voxels;
    float scale;
    const int *num_wins;  // device
    const int *win_ind;   // (cap,4) [b,wz,wy,wx]
    const int *indices;   // (N,4)  [b,z,y,x]
    const int *k_ind, *win_vstart, *win_cnt;  // K4 lists
    const int *pair_win;                      // (N) window of every voxel, -1: in no list
    float vsx, vsy, vsz, minx, miny, minz, wsx, wsy, wsz;
    const float *xhat;
    const float *Wp1, *bp1, *Wp2, *bp2;  // pos_proj.0 (C,6), pos_proj.2 (C,C)
    const float *Wq, *bq, *Wkv, *bkv, *Wo, *bo;
    float *qp, *ktok, *score, *vp, *out;
};

__device__ __forceinline__ float cf_centre(int idx, float cell, float lo) {
    return __fadd_rn(__fmul_rn(__fadd_rn((float)idx, 0.5f), cell), lo);  // ref with_coords :132-137
}

// rows x COLS matrix -> LDS rows of LS floats, 8 float4 in flight per thread
template <int COLS, int LS>
__device__ __forceinline__ void cf_stage(float *dst, const float *src, int rows) {
    constexpr int UN = 8;
    const int total = rows * COLS;
    for (int e0 = threadIdx.x * 4; e0 < total; e0 += blockDim.x * 4 * UN) {
        float4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int e = e0 + u * blockDim.x * 4;
            v[u] = e < total ? *reinterpret_cast<const float4 *>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int e = e0 + u * blockDim.x * 4;
            if (e < total) *reinterpret_cast<float4 *>(dst + (e / COLS) * LS + e % COLS) = v[u];
        }
    }
}

// D^T[out][row] += W[out][:] . x[row][:] for NT x NT tiles; acc[u] must hold the bias
template <int NT, int LS>
__device__ __forceinline__ void cf_gemm(f32x4 (&acc)[NT], const float *W_l, const f32x4 (&x)[NT], int la, int g) {
    constexpr int UG = NT > 4 ? 4 : NT;  // output tiles per group = independent MFMA chains
#pragma unroll
    for (int u0 = 0; u0 < NT; u0 += UG) {
        const float *wbase = W_l + (size_t)(16 * u0 + la) * LS + 4 * g;
#pragma unroll
        for (int S = 0; S < NT; ++S) {
            float4 w[UG];
#pragma unroll
            for (int u = 0; u < UG; ++u) w[u] = *reinterpret_cast<const float4 *>(wbase + u * 16 * LS + 16 * S);
#pragma unroll
            for (int u = 0; u < UG; ++u) MFMA4(acc[u0 + u], w[u].x, x[S][0]);
#pragma unroll
            for (int u = 0; u < UG; ++u) MFMA4(acc[u0 + u], w[u].y, x[S][1]);
#pragma unroll
            for (int u = 0; u < UG; ++u) MFMA4(acc[u0 + u], w[u].z, x[S][2]);
#pragma unroll
            for (int u = 0; u < UG; ++u) MFMA4(acc[u0 + u], w[u].w, x[S][3]);
            __builtin_amdgcn_sched_barrier(0);  // keep the ds_reads of later steps where they are (VGPRs)
        }
    }
}

// ---- the same products with every fp32 operand split into two fp16 halves (22 of 24 mantissa bits) (see ffn.hip, k_ffn_ws): hi =
// fp16(v), lo = fp16((v - hi) 2^11); sum a b = sum a_hi b_hi + 2^-11 (sum a_hi b_lo + sum a_lo b_hi), three
// v_mfma_f32_16x16x32_f16 with fp32 accumulation: the fp32 instruction's error against float64 at 3/16 of its cycles.
// LDS row of a weight matrix: C halves hi | C halves lo (+ 16 bytes: the same (C + 4)-float stride, conflict free);
// inside a 32-channel group channel 16 h + 4 g + j sits at half 8 g + 4 h + j, so that lane (row, g) reads its 8 k
// slots as one ds_read_b128 and the B operand is just the split of x[2 P] | x[2 P + 1] (lane (row, g): channels
// 32 P + 4 g + j and 32 P + 16 + 4 g + j).  The caller guarantees the fp16 range of the operands (fused.py).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
#define CF_SCALE 2048.0f
#define CF_INV (1.0f / 2048.0f)

__device__ __forceinline__ void cf_split4(const float v0, const float v1, const float v2, const float v3, h16x4 &hi, h16x4 &lo) {
    const fp16x2 a = __builtin_amdgcn_cvt_pkrtz(v0, v1), b = __builtin_amdgcn_cvt_pkrtz(v2, v3);
    // (v - hi) 2^11 as fma(hi, -2^11, v 2^11): exact steps, same bits, hi converted inside v_fma_mix_f32
    const fp16x2 c = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)a[0], -CF_SCALE, v0 * CF_SCALE), __builtin_fmaf((float)a[1], -CF_SCALE, v1 * CF_SCALE));
    const fp16x2 d = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)b[0], -CF_SCALE, v2 * CF_SCALE), __builtin_fmaf((float)b[1], -CF_SCALE, v3 * CF_SCALE));
    hi = h16x4{(_Float16)a[0], (_Float16)a[1], (_Float16)b[0], (_Float16)b[1]};
    lo = h16x4{(_Float16)c[0], (_Float16)c[1], (_Float16)d[0], (_Float16)d[1]};
}

template <int COLS, int LS>
__device__ __forceinline__ void cf_stage_h(float *dst, const float *src, int rows) {
    constexpr int UN = 8;
    const int total = rows * COLS;
    for (int e0 = threadIdx.x * 4; e0 < total; e0 += blockDim.x * 4 * UN) {
        float4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int e = e0 + u * blockDim.x * 4;
            v[u] = e < total ? *reinterpret_cast<const float4 *>(src + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int e = e0 + u * blockDim.x * 4;
            if (e < total) {
                const int c = e % COLS, c32 = c & 31;  // c32 = 16 h + 4 g (a float4 never straddles a group of 4)
                const int half = (c - c32) + 8 * ((c32 >> 2) & 3) + 4 * (c32 >> 4);
                h16x4 hi, lo;
                cf_split4(v[u].x, v[u].y, v[u].z, v[u].w, hi, lo);
                _Float16 *row = reinterpret_cast<_Float16 *>(dst + (size_t)(e / COLS) * LS);
                *reinterpret_cast<h16x4 *>(row + half) = hi;
                *reinterpret_cast<h16x4 *>(row + COLS + half) = lo;
            }
        }
    }
}

template <int NT, int LS>
__device__ __forceinline__ void cf_gemm_h(f32x4 (&acc)[NT], const float *W_l, const f32x4 (&x)[NT], int la, int g) {
    static_assert(NT % 2 == 0, "32-channel k steps");
    constexpr int NP = NT / 2, UG = NT > 4 ? 4 : NT;
    h16x8 xh[NP], xl[NP];
#pragma unroll
    for (int P = 0; P < NP; ++P) {
        h16x4 h0, l0, h1, l1;
        cf_split4(x[2 * P][0], x[2 * P][1], x[2 * P][2], x[2 * P][3], h0, l0);
        cf_split4(x[2 * P + 1][0], x[2 * P + 1][1], x[2 * P + 1][2], x[2 * P + 1][3], h1, l1);
        xh[P] = h16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        xl[P] = h16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    }
#pragma unroll
    for (int u0 = 0; u0 < NT; u0 += UG) {
        const _Float16 *wbase = reinterpret_cast<const _Float16 *>(W_l + (size_t)(16 * u0 + la) * LS) + 8 * g;
        f32x4 lo[UG];
#pragma unroll
        for (int u = 0; u < UG; ++u) lo[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int P = 0; P < NP; ++P) {
            h16x8 wh[UG], wl[UG];
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const _Float16 *wr = wbase + (size_t)u * 16 * LS * 2 + 32 * P;
                wh[u] = *reinterpret_cast<const h16x8 *>(wr);
                wl[u] = *reinterpret_cast<const h16x8 *>(wr + 16 * NT);
            }
#pragma unroll
            for (int u = 0; u < UG; ++u) acc[u0 + u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u], xh[P], acc[u0 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < UG; ++u) lo[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[u], xl[P], lo[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < UG; ++u) lo[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[u], xh[P], lo[u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < UG; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[u0 + u][i] = __builtin_fmaf(lo[u][i], CF_INV, acc[u0 + u][i]);
    }
}

template <bool H, int COLS, int LS>
__device__ __forceinline__ void cf_stage_x(float *dst, const float *src, int rows) {
    if (H) cf_stage_h<COLS, LS>(dst, src, rows);
    else cf_stage<COLS, LS>(dst, src, rows);
}
template <bool H, int NT, int LS>
__device__ __forceinline__ void cf_gemm_x(f32x4 (&acc)[NT], const float *W_l, const f32x4 (&x)[NT], int la, int g) {
    if (H) cf_gemm_h<NT, LS>(acc, W_l, x, la, g);
    else cf_gemm<NT, LS>(acc, W_l, x, la, g);
}

// The K4 lists of a tile's 16 windows -> LDS (one coalesced round trip instead of one dependent
// load per list step); lst[r * ns + s] = global feature row of slot s of window r (or of slot 0 when
// s >= cnt: a valid row whose contribution is masked by the caller)
__device__ __forceinline__ void cf_stage_lists(int *lst, const CmpArgs &a, int tile, int nw, int lane) {
    const int total = 16 * a.ns;
    for (int e = lane; e < total; e += MSSVT_WAVE) {
        const int r = e / a.ns, sl = e % a.ns;
        const int w = min(tile * 16 + r, nw - 1);
        const int cnt = a.win_cnt[w];
        // (a window whose voxels were all dropped by an overflowing hash table has cnt = 0 and slot 0 = -1: the
        // frame ends in an error anyway, but every row read on the way must exist)
        lst[e] = a.win_vstart[w] + max(a.k_ind[(size_t)w * a.ns + (sl < cnt ? sl : 0)], 0);
    }
    wave_lds_sync();
}

// ---- A: queries ---------------------------------------------------------------------------------
template <int C, bool H>
__device__ __forceinline__ void cmp_query_body(const CmpArgs &a, int block_id, int num_blocks) {
    constexpr int NT = C / 16, LS = C + 4;
    extern __shared__ float4 lds4[];
    float *Wq_l = reinterpret_cast<float *>(lds4), *bq_l = Wq_l + C * LS;
    int *lst = reinterpret_cast<int *>(bq_l + C) + (threadIdx.x / MSSVT_WAVE) * 16 * a.ns;
    cf_stage_x<H, C, LS>(Wq_l, a.Wq, C);
    for (int e = threadIdx.x; e < C; e += blockDim.x) bq_l[e] = a.bq[e];
    __syncthreads();
    const int lane = lane_id(), la = lane & 15, g = lane >> 4, wv = threadIdx.x / MSSVT_WAVE;
    const int nw = *a.num_wins, tiles = (nw + 15) >> 4;
    for (int k = wv;; k += CFQ_NW) {
        const int tile = k * num_blocks + block_id;
        if (tile >= tiles) break;
        const int w = min(tile * 16 + la, nw - 1);
        const bool live = tile * 16 + la < nw;
        const int cnt = a.win_cnt[w];
        cf_stage_lists(lst, a, tile, nw, lane);
        const int *list = lst + la * a.ns;
        // q_tok = max over the zero padded key features: empty slots contribute zeros (ref :370)
        f32x4 m[NT];
        const float init = cnt < a.ns ? 0.0f : -INFINITY;
#pragma unroll
        for (int S = 0; S < NT; ++S) m[S] = f32x4{init, init, init, init};
        int cmax = cnt;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) cmax = max(cmax, __shfl_xor(cmax, off));
        // two list steps per iteration: 2 x NT row loads in flight (slots >= cnt re-read slot 0: max unchanged)
        for (int s = 0; s < cmax; s += 2) {
            const float *x0 = a.xhat + (size_t)list[s] * C + 4 * g;
            const float *x1 = a.xhat + (size_t)list[min(s + 1, a.ns - 1)] * C + 4 * g;
            float4 v0[NT], v1[NT];
#pragma unroll
            for (int S = 0; S < NT; ++S) {
                v0[S] = *reinterpret_cast<const float4 *>(x0 + 16 * S);
                v1[S] = *reinterpret_cast<const float4 *>(x1 + 16 * S);
            }
#pragma unroll
            for (int S = 0; S < NT; ++S) {
                m[S][0] = fmaxf(m[S][0], fmaxf(v0[S].x, v1[S].x));
                m[S][1] = fmaxf(m[S][1], fmaxf(v0[S].y, v1[S].y));
                m[S][2] = fmaxf(m[S][2], fmaxf(v0[S].z, v1[S].z));
                m[S][3] = fmaxf(m[S][3], fmaxf(v0[S].w, v1[S].w));
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float4 b = *reinterpret_cast<const float4 *>(bq_l + 16 * u + 4 * g);
            acc[u] = f32x4{b.x, b.y, b.z, b.w};
        }
        cf_gemm_x<H, NT, LS>(acc, Wq_l, m, la, g);
        // the first read of the sums in straight-line code BEHIND the chain, not inside the predicated store region
        // (tools/mfma_hazard_check.py: a wave that skips the region reaches the next tile's register writes early)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[u] = acc[u] * a.scale;
        if (live) {
            float *dst = a.qp + (size_t)w * C + 4 * g;
#pragma unroll
            for (int u = 0; u < NT; ++u)
                *reinterpret_cast<float4 *>(dst + 16 * u) = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
        }
    }
}

// ---- B: key tokens --------------------------------------------------------------------------------
template <int C, bool H>
__device__ __forceinline__ void cmp_keys_body(const CmpArgs &a, int block_id, int num_blocks) {
    constexpr int NT = C / 16, LS = C + 4;
    extern __shared__ float4 lds4[];
    float *W2_l = reinterpret_cast<float *>(lds4), *b2_l = W2_l + C * LS;
    cf_stage_x<H, C, LS>(W2_l, a.Wp2, C);
    for (int e = threadIdx.x; e < C; e += blockDim.x) b2_l[e] = a.bp2[e];
    const int lane = lane_id(), la = lane & 15, g = lane >> 4, wv = threadIdx.x / MSSVT_WAVE;
    // layer 1 as a K = 8 product: inputs d = 4 s + g of step s = (rel.x, rel.y, rel.z, c.x | c.y, c.z, 1, 0);
    // A operand of this lane: row = channel 16 t + la of pos_proj.0 extended by its bias
    float w1a[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float *wr = a.Wp1 + (size_t)(16 * t + la) * 6;
        w1a[t][0] = wr[g];
        w1a[t][1] = lane_pick4(g, wr[4], wr[5], a.bp1[16 * t + la], 0.f);
    }
    __syncthreads();
    const int n = a.num_voxels, tiles = (n + 15) >> 4;
    for (int k = wv;; k += CFQ_NW) {
        const int tile = k * num_blocks + block_id;
        if (tile >= tiles) break;
        const int v = min(tile * 16 + la, n - 1);
        const bool live = tile * 16 + la < n;
        const int pw = a.pair_win[v];
        const int4 vi = reinterpret_cast<const int4 *>(a.indices)[v];
        const int4 wi = reinterpret_cast<const int4 *>(a.win_ind)[max(pw, 0)];
        const float cxm = cf_centre(wi.w, a.wsx, a.minx), cym = cf_centre(wi.z, a.wsy, a.miny),
                    czm = cf_centre(wi.y, a.wsz, a.minz);
        const float rx = cf_centre(vi.w, a.vsx, a.minx) - cxm, ry = cf_centre(vi.z, a.vsy, a.miny) - cym,
                    rz = cf_centre(vi.y, a.vsz, a.minz) - czm;  // NOT masked in the CompressBlock (ref :372)
        const float in0 = lane_pick4(g, rx, ry, rz, cxm);
        const float in1 = lane_pick4(g, cym, czm, 1.0f, 0.0f);
        const float *xr = a.xhat + (size_t)v * C + 4 * g;
        float4 xv[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) xv[u] = *reinterpret_cast<const float4 *>(xr + 16 * u);
        f32x4 h[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 p = f32x4{0.f, 0.f, 0.f, 0.f};
            MFMA4(p, w1a[t][0], in0);
            MFMA4(p, w1a[t][1], in1);
            h[t] = f32x4{fmaxf(p[0], 0.f), fmaxf(p[1], 0.f), fmaxf(p[2], 0.f), fmaxf(p[3], 0.f)};
        }
        f32x4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float4 b = *reinterpret_cast<const float4 *>(b2_l + 16 * u + 4 * g);
            acc[u] = f32x4{b.x, b.y, b.z, b.w};
        }
        cf_gemm_x<H, NT, LS>(acc, W2_l, h, la, g);
        if (live) {
            float *dst = a.ktok + (size_t)v * C + 4 * g;
            const bool in_list = pw >= 0;  // a voxel in no list (truncated window): finite dummy row
#pragma unroll
            for (int u = 0; u < NT; ++u)
                *reinterpret_cast<float4 *>(dst + 16 * u) =
                    in_list ? make_float4(xv[u].x + fmaxf(acc[u][0], 0.f), xv[u].y + fmaxf(acc[u][1], 0.f),
                                          xv[u].z + fmaxf(acc[u][2], 0.f), xv[u].w + fmaxf(acc[u][3], 0.f))
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// A and B in ONE launch: they are independent (windows -> qp, voxels -> ktok), and the query side is a
// latency-bound walk over the window lists that hides under the key side's matrix work.
template <int C, bool H>
__global__ void __launch_bounds__(CFQ_NW *MSSVT_WAVE) k_cmp_query_keys(CmpArgs a, int query_blocks) {
    // roles alternate so that both kinds are resident from the start (dispatch is in block order)
    const int key_blocks = gridDim.x - query_blocks, both = 2 * min(query_blocks, key_blocks);
    const int b = blockIdx.x;
    const bool query = b < both ? (b & 1) == 0 : query_blocks > key_blocks;
    const int id = b < both ? b >> 1 : b - both / 2;
    if (query)
        cmp_query_body<C, H>(a, id, query_blocks);
    else
        cmp_keys_body<C, H>(a, id, key_blocks);
}

// ---- C: K scores + V rows ---------------------------------------------------------------------------
template <int C, int HD, bool H>
__global__ void __launch_bounds__(CFK_NW *MSSVT_WAVE) k_cmp_kv(CmpArgs a) {
    constexpr int NT = C / 16, LS = C + 4, NH = C / HD;
    static_assert(HD == 8 || HD == 16 || HD == 32, "head dims instantiated: 8, 16, 32");
    extern __shared__ float4 lds4[];
    float *W_l = reinterpret_cast<float *>(lds4), *b_l = W_l + 2 * C * LS;  // [Wk ; Wv] rows, [bk ; bv]
    cf_stage_x<H, C, LS>(W_l, a.Wkv, 2 * C);
    for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) b_l[e] = a.bkv[e];
    __syncthreads();
    const int lane = lane_id(), la = lane & 15, g = lane >> 4, wv = threadIdx.x / MSSVT_WAVE;
    const int n = a.num_voxels, tiles = (n + 15) >> 4;
    for (int k = wv;; k += CFK_NW) {
        const int tile = k * gridDim.x + blockIdx.x;
        if (tile >= tiles) break;
        const int v = min(tile * 16 + la, n - 1);
        const bool live = tile * 16 + la < n;
        const int pw = a.pair_win[v];
        const float *xr = a.ktok + (size_t)v * C + 4 * g;
        const float *qr = a.qp + (size_t)max(pw, 0) * C + 4 * g;
        f32x4 x[NT];
        float4 q[NT];
#pragma unroll
        for (int S = 0; S < NT; ++S) {
            const float4 t4 = *reinterpret_cast<const float4 *>(xr + 16 * S);
            x[S] = f32x4{t4.x, t4.y, t4.z, t4.w};
            q[S] = *reinterpret_cast<const float4 *>(qr + 16 * S);
        }
        // K = Wk k_tok + bk, reduced against the window's (pre-scaled) query right away
        f32x4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float4 b = *reinterpret_cast<const float4 *>(b_l + 16 * u + 4 * g);
            acc[u] = f32x4{b.x, b.y, b.z, b.w};
        }
        cf_gemm_x<H, NT, LS>(acc, W_l, x, la, g);
        float part[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            part[u] = (acc[u][0] * q[u].x + acc[u][1] * q[u].y) + (acc[u][2] * q[u].z + acc[u][3] * q[u].w);
            // this lane's 4 channels 16 u + 4 g .. belong to head (16 u + 4 g) / HD
            if (HD >= 16) {
                part[u] += lane_xor16(part[u]);
                part[u] += lane_xor32(part[u]);
            } else {  // HD == 8: lanes g = 0,1 -> head 2u, g = 2,3 -> head 2u + 1
                part[u] += lane_xor16(part[u]);
            }
        }
        if (live && pw >= 0) {
            float *sd = a.score + (size_t)v * NH;
            if (HD == 16) {
                if (g == 0) {
#pragma unroll
                    for (int u = 0; u < NT; ++u) sd[u] = part[u];
                }
            } else if (HD == 32) {
                if (g == 0) {
#pragma unroll
                    for (int u = 0; u < NT; u += 2) sd[u / 2] = part[u] + part[u + 1];
                }
            } else {
                if ((g & 1) == 0) {
#pragma unroll
                    for (int u = 0; u < NT; ++u) sd[2 * u + (g >> 1)] = part[u];
                }
            }
        }
        // V = Wv k_tok + bv
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float4 b = *reinterpret_cast<const float4 *>(b_l + C + 16 * u + 4 * g);
            acc[u] = f32x4{b.x, b.y, b.z, b.w};
        }
        cf_gemm_x<H, NT, LS>(acc, W_l + (size_t)C * LS, x, la, g);
        if (live) {
            float *dst = a.vp + (size_t)v * C + 4 * g;
#pragma unroll
            for (int u = 0; u < NT; ++u)
                *reinterpret_cast<float4 *>(dst + 16 * u) = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
        }
    }
}

// ---- D: softmax, weighted V sum, output projection --------------------------------------------------
template <int C, int HD, bool H>
__global__ void __launch_bounds__(CFO_NW *MSSVT_WAVE) k_cmp_out(CmpArgs a) {
    constexpr int NT = C / 16, LS = C + 4, NH = C / HD;
    extern __shared__ float4 lds4[];
    float *Wo_l = reinterpret_cast<float *>(lds4), *bo_l = Wo_l + C * LS;
    int *lst = reinterpret_cast<int *>(bo_l + C) + (threadIdx.x / MSSVT_WAVE) * 16 * a.ns;
    cf_stage_x<H, C, LS>(Wo_l, a.Wo, C);
    for (int e = threadIdx.x; e < C; e += blockDim.x) bo_l[e] = a.bo[e];
    __syncthreads();
    const int lane = lane_id(), la = lane & 15, g = lane >> 4, wv = threadIdx.x / MSSVT_WAVE;
    const int nw = *a.num_wins, tiles = (nw + 15) >> 4;
    for (int k = wv;; k += CFO_NW) {
        const int tile = k * gridDim.x + blockIdx.x;
        if (tile >= tiles) break;
        const int w = min(tile * 16 + la, nw - 1);
        const bool live = tile * 16 + la < nw;
        const int cnt = a.win_cnt[w];
        cf_stage_lists(lst, a, tile, nw, lane);
        const int *list = lst + la * a.ns;
        int cmax = cnt;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) cmax = max(cmax, __shfl_xor(cmax, off));
        // online softmax over the window's voxels, one pass; this lane's channel group of tile S belongs
        // to head (16 S + 4 g) / HD
        float mx[NT], sum[NT];
        f32x4 o[NT];
#pragma unroll
        for (int S = 0; S < NT; ++S) {
            mx[S] = -INFINITY;
            sum[S] = 0.f;
            o[S] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int s = 0; s < cmax; ++s) {
            const bool on = s < cnt;
            const int vrow = list[s];
            const float *sr = a.score + (size_t)vrow * NH;
            const float *vr = a.vp + (size_t)vrow * C + 4 * g;
            float4 vv[NT];
            float sc[NT];
#pragma unroll
            for (int S = 0; S < NT; ++S) {
                vv[S] = *reinterpret_cast<const float4 *>(vr + 16 * S);
                sc[S] = sr[(16 * S + 4 * g) / HD];
            }
            if (on) {
#pragma unroll
                for (int S = 0; S < NT; ++S) {
                    const float mn = fmaxf(mx[S], sc[S]);
                    const float corr = __expf(mx[S] - mn), e = __expf(sc[S] - mn);  // first step: exp(-inf) = 0
                    mx[S] = mn;
                    sum[S] = sum[S] * corr + e;
                    o[S][0] = __builtin_fmaf(e, vv[S].x, o[S][0] * corr);
                    o[S][1] = __builtin_fmaf(e, vv[S].y, o[S][1] * corr);
                    o[S][2] = __builtin_fmaf(e, vv[S].z, o[S][2] * corr);
                    o[S][3] = __builtin_fmaf(e, vv[S].w, o[S][3] * corr);
                }
            }
        }
#pragma unroll
        for (int S = 0; S < NT; ++S) {
            // every window owns >= 1 listed voxel -- except one made by a voxel outside the grid (invalid input: the
            // voxel table skips it, K2 does not): the reference would average its padded slots, here the row is bo
            const float inv = sum[S] > 0.f ? 1.0f / sum[S] : 0.f;
            o[S][0] *= inv; o[S][1] *= inv; o[S][2] *= inv; o[S][3] *= inv;
        }
        f32x4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        cf_gemm_x<H, NT, LS>(acc, Wo_l, o, la, g);
        // bias BEHIND the product: the first read of the sums is straight-line code, not the predicated store region
        // (tools/mfma_hazard_check.py: a wave that skips the region reaches the next tile's register writes early)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float4 b = *reinterpret_cast<const float4 *>(bo_l + 16 * u + 4 * g);
            acc[u] = acc[u] + f32x4{b.x, b.y, b.z, b.w};
        }
        if (live) {
            float *dst = a.out + (size_t)w * C + 4 * g;
#pragma unroll
            for (int u = 0; u < NT; ++u)
                *reinterpret_cast<float4 *>(dst + 16 * u) = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
        }
    }
}

template <typename K>
static int cf_prepare(K kernel, size_t lds_bytes) {
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    return MSSVT_OK;
}

template <int C, int HD, bool H>
static int launch_compress(const CmpArgs &a, int win_capacity, hipStream_t stream) {
    constexpr int LS = C + 4;
    const size_t lds1 = ((size_t)C * LS + C) * 4, lds2 = ((size_t)2 * C * LS + 2 * C) * 4;
    // + the tile's K4 lists per wave
    const size_t lds_q = lds1 + (size_t)CFQ_NW * 16 * a.ns * 4, lds_o = lds1 + (size_t)CFO_NW * 16 * a.ns * 4;
    if (lds_q > 160 * 1024 || lds_o > 160 * 1024) return MSSVT_E_TOOLARGE;
    int rc;
    if ((rc = cf_prepare(k_cmp_query_keys<C, H>, lds_q)) ||
        (rc = cf_prepare(k_cmp_kv<C, HD, H>, lds2)) || (rc = cf_prepare(k_cmp_out<C, HD, H>, lds_o)))
        return rc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int vt = (a.num_voxels + 15) / 16, wt = (win_capacity + 15) / 16;
    // every workgroup resident at once (blocks per CU by LDS); the roles of the first launch share them
    const int per_q = (int)((160 * 1024) / lds_q) < 1 ? 1 : (int)((160 * 1024) / lds_q);
    const int per_o = (int)((160 * 1024) / lds_o) < 1 ? 1 : (int)((160 * 1024) / lds_o);
    const int slots = cus * per_q;
    // fp32 products: ~1 : 2 (a key tile carries a C x C product on top of its positional layer, a query tile mostly waits
    // for its list walk); with the split-fp16 products the key side is cheap and the latency-bound query side sets the
    // launch time: 1 : 1 measured best (33 % -> 42 / 319 us at one / eight scenes, 50 % -> 36 / 230, 67 % -> 44 / 309)
    int g_w = min(max(H ? slots / 2 : slots / 3, 1), max(wt, 1)), g_v = min(max(slots - g_w, 1), max(vt, 1));
    const int g_v2 = min(cus, max(vt, 1)), g_o = min(cus * per_o, max(wt, 1));
    k_cmp_query_keys<C, H><<<g_w + g_v, CFQ_NW * MSSVT_WAVE, lds_q, stream>>>(a, g_w);
    k_cmp_kv<C, HD, H><<<g_v2, CFK_NW * MSSVT_WAVE, lds2, stream>>>(a);
    k_cmp_out<C, HD, H><<<g_o, CFO_NW * MSSVT_WAVE, lds_o, stream>>>(a);
    return mssvt_launch_status();
}

extern "C" int mssvt_compress_fused(
    int C, int head_dim, float scale, int max_num_win1, int num_voxels, const int *num_wins_dev, int win_capacity,
    const int *win_ind, const int *indices, const int *k_ind, const int *win_vstart, const int *win_cnt,
    const int *pair_win, const float *host_voxel_size3, const float *host_range_min3, const float *host_win_size3,
    const float *xhat, const float *Wpos1, const float *bpos1, const float *Wpos2, const float *bpos2,
    const float *Wq, const float *bq, const float *Wkv, const float *bkv, const float *Wo, const float *bo,
    float *qp, float *ktok, float *score, float *vp, float *out, int split_f16, void *stream) {
    if (!num_wins_dev || !win_ind || !indices || !k_ind || !win_vstart || !win_cnt || !pair_win ||
        !host_voxel_size3 || !host_range_min3 || !host_win_size3 || !xhat || !Wpos1 || !bpos1 || !Wpos2 || !bpos2 ||
        !Wq || !bq || !Wkv || !bkv || !Wo || !bo || !qp || !ktok || !score || !vp || !out || C <= 0 ||
        head_dim <= 0 || max_num_win1 <= 0 || num_voxels < 0 || win_capacity <= 0)
        return MSSVT_E_BADARG;
    if (C % head_dim) return MSSVT_E_BADARG;
    if (num_voxels == 0) return MSSVT_OK;
    CmpArgs a;
    a.C = C; a.ns = max_num_win1; a.num_voxels = num_voxels; a.scale = scale;
    a.num_wins = num_wins_dev; a.win_ind = win_ind; a.indices = indices;
    a.k_ind = k_ind; a.win_vstart = win_vstart; a.win_cnt = win_cnt; a.pair_win = pair_win;
    a.vsx = host_voxel_size3[0]; a.vsy = host_voxel_size3[1]; a.vsz = host_voxel_size3[2];
    a.minx = host_range_min3[0]; a.miny = host_range_min3[1]; a.minz = host_range_min3[2];
    a.wsx = host_win_size3[0]; a.wsy = host_win_size3[1]; a.wsz = host_win_size3[2];
    a.xhat = xhat; a.Wp1 = Wpos1; a.bp1 = bpos1; a.Wp2 = Wpos2; a.bp2 = bpos2;
    a.Wq = Wq; a.bq = bq; a.Wkv = Wkv; a.bkv = bkv; a.Wo = Wo; a.bo = bo;
    a.qp = qp; a.ktok = ktok; a.score = score; a.vp = vp; a.out = out;
    hipStream_t st = (hipStream_t)stream;
#define CF_CASE(c, hd)                                                                                    \
    if (C == c && head_dim == hd)                                                                         \
        return split_f16 ? launch_compress<c, hd, true>(a, win_capacity, st) : launch_compress<c, hd, false>(a, win_capacity, st);
    CF_CASE(128, 16)
    CF_CASE(128, 32)
    CF_CASE(64, 8)
    CF_CASE(64, 16)
    CF_CASE(64, 32)
    CF_CASE(32, 8)
    CF_CASE(32, 16)
    CF_CASE(32, 32)
#undef CF_CASE
    return MSSVT_E_TOOLARGE;  // shape not instantiated: the caller uses the ragged kernels of compress.hip
}
