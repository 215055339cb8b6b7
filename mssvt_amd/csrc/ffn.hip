// ffn.hip -- fused per-voxel feed-forward tail of an MsSVT block on the matrix cores.
//
// Two generations live here (the first one -- one launch, both weight matrices streamed through the LDS in chunks of 32
// hidden units, fp32 MFMA, ~40 % of the fp32 matrix peak -- was deleted in round 3; DESIGN.md section 5.1 keeps its numbers):
//   k_ffn_up/down  two launches, whole weight matrix resident in LDS, fp32 MFMA             (ffn_arith = "f32", and
//                                                                      parameters outside the fp16 range)
//   k_ffn_ws       one launch, weights stationary in registers, fp32 operands split into two fp16 halves
//                  (3 x v_mfma_f32_16x16x32_f16 per product sum: fp32-instruction accuracy)   (the default)
//
// Replaces, per block, the reference's (ref: mssvt_backbone.py:298-343 / :383-387)
//     3-NN interpolation + scatter + shortcut      K9, K10, index_put, add
//     h   = norm2(x)                                LayerNorm kernel
//     u   = relu(linear1(h))                        GEMM (N x C x FF) + bias + ReLU kernels
//     y   = x + linear2(u)                          GEMM (N x FF x C) + bias + add kernels
// and the NEXT block's norm1(y): x is built while it is loaded, y (and optionally norm1_next(y)) written once.
//
// HBM access is fully coalesced although the MFMA operand layouts are not: rows are loaded and stored as whole rows
// (float4 per lane) and re-laid out on chip.
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));


struct FfnArgs {
    int n_rows;
    const int *n_rows_dev;  // optional: the row count lives on the device (n_rows = capacity)
    const float *x_new, *x_in;  // plain input: x = owner && owner[v] < 0 ? 2 * x_in[v] : x_new[v]
    const int *owner;
    // interpolation-table input (tab_row != null):
    //   x = tab_row[v].x < 0 ? 2 * x_in[v] : x_in[v] + sum_i tab_w[v][i] * attn[tab_row[v][i]]
    const int4 *tab_row;
    const float4 *tab_w;
    const float *attn;
    const float *ln_w, *ln_b;  // norm2
    float eps;
    const float *W1, *b1, *W2, *b2;  // linear1 (FF,C), linear2 (C,FF)
    float *y;
    const float *ln2_w, *ln2_b;  // optional second LayerNorm applied to y (next block's norm1)
    float eps2;
    float *y_norm;
    int xcd;  // deal the tiles so that an XCD owns a contiguous run per round (common.hip.h, xcd_contiguous_block)
    int prio;
};

// ---------------------------------------------------------------------------------------------
// Split form: two launches with the WHOLE weight matrix of each GEMM resident in LDS
// (W1: FF x (C+4) floats = 132 KiB at C=128, FF=256; W2: C x (FF+4) = 130 KiB), one 16-wave
// workgroup per CU, 16 rows per wavefront, NO barrier after the weights are staged: waves run
// independently, so one wave's row gathers / LayerNorm / stores overlap with the MFMAs of the
// other three on its SIMD.  (The deleted single-launch form streamed both matrices through LDS for every
// 64 rows, 8 barriers per batch, ~40 % of the fp32 MFMA peak.)  The price of the split
// is one (N, FF) round trip of the hidden activations (4*FF bytes per row written + read).
// Both GEMMs are computed transposed (D^T[out][row]), see block_attn.hip: A = weight rows (one
// ds_read_b128 per 4 k-steps, conflict free with the +4 padding), B = the activations held as
// lane (row = l % 16, g = l / 16) -> channels 16 S + 4 g + j, i.e. plain 16-byte global loads; the
// accumulator (lane (row, g), register i = output 16 t + 4 g + i) leaves as 16-byte stores.
// ---------------------------------------------------------------------------------------------
#ifdef MSSVT_STAMPS  // developer instrumentation: per-wave s_memtime stamps of the split kernels
__device__ unsigned long long g_stamps[2 * 4 * 8 * 64];
extern "C" int mssvt_debug_read_stamps(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#define STAMP(kid_) \
    if (lane == 0 && blockIdx.x < 4 && si < 64) g_stamps[(((kid_) * 4 + blockIdx.x) * 8 + wv) * 64 + si++] = __builtin_readcyclecounter();
#else
#define STAMP(kid_)
#endif
#define FFS_NW 8  // waves per workgroup of the split kernels (2 per SIMD, up to 256 VGPRs each)

#define MFMA4(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av), (bv), acc, 0, 0, 0)

// Residual input of one row: x = tab.x < 0 ? 2 x_in : x_in + sum_i w_i attn[row_i]  (table form), or
// x_new / 2 x_in by owner.  Four row pointers + weights; every stream is read with the same pattern.
struct FfnRowSrc {
    const float *px, *p1, *p2, *p3;
    float w1, w2, w3, wx;
    bool live;
    float *py, *pu;
};

// table entry of a row (loaded one tile ahead of the gathers that need it)
struct FfnTab {
    int4 tr;
    float4 tw;
    int own;
};

__device__ __forceinline__ FfnTab ffn_tab_load(const FfnArgs &a, int tile, int n, int la) {
    FfnTab t;
    const int row = min(tile * 16 + la, n - 1);
    t.tr = make_int4(0, 0, 0, 0);
    t.tw = make_float4(0.f, 0.f, 0.f, 0.f);
    t.own = 0;
    if (a.tab_row) {
        t.tr = a.tab_row[row];
        t.tw = a.tab_w[row];
    } else if (a.owner) {
        t.own = a.owner[row];
    }
    return t;
}

template <int C, int FF>
__device__ __forceinline__ FfnRowSrc ffn_row_src(const FfnArgs &a, float *hidden, int tile, int n, int la, int g,
                                                 const FfnTab &t) {
    FfnRowSrc r;
    const int row = min(tile * 16 + la, n - 1);
    r.live = tile * 16 + la < n;
    r.py = a.y + (size_t)row * C + 4 * g;
    r.pu = hidden + (size_t)row * FF + 4 * g;
    if (a.tab_row) {
        const bool unowned = t.tr.x < 0;
        // an unowned voxel re-reads its own (finite) x_in row with weight 0: attention rows of
        // never-written slots may hold NaNs, and 0 * NaN is NaN
        r.px = a.x_in + (size_t)row * C + 4 * g;
        r.p1 = unowned ? r.px : a.attn + (size_t)t.tr.x * C + 4 * g;
        r.p2 = unowned ? r.px : a.attn + (size_t)t.tr.y * C + 4 * g;
        r.p3 = unowned ? r.px : a.attn + (size_t)t.tr.z * C + 4 * g;
        r.w1 = unowned ? 0.f : t.tw.x;
        r.w2 = unowned ? 0.f : t.tw.y;
        r.w3 = unowned ? 0.f : t.tw.z;
        r.wx = unowned ? 2.0f : 1.0f;  // untouched voxel: features + shortcut = 2 * x_in
    } else {
        const bool dbl = a.owner != nullptr && t.own < 0;  // untouched voxel (ref quirk R12)
        r.px = r.p1 = r.p2 = r.p3 = (dbl ? a.x_in : a.x_new) + (size_t)row * C + 4 * g;
        r.w1 = r.w2 = r.w3 = 0.f;
        r.wx = dbl ? 2.0f : 1.0f;
    }
    return r;
}

// weights -> LDS rows of LS floats, UN float4 loads in flight per thread (a plain copy loop waits
// for every load before its store: 17 round trips for 132 KiB; UN = 16 at 512 threads is ONE)
template <int COLS, int LS, int UN = 16>
__device__ __forceinline__ void ffn_stage_weights(float *dst, const float *src, int rows) {
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

// macros, not lambdas: handing register arrays to a lambda takes their address -> scratch memory
#define FFN_ISSUE(src_, S0_)                                                                     \
    _Pragma("unroll") for (int s_ = 0; s_ < SH; ++s_) {                                          \
        rx[s_] = *reinterpret_cast<const float4 *>((src_).px + 16 * ((S0_) + s_));              \
        if (tabbed) {                                                                            \
            r1[s_] = *reinterpret_cast<const float4 *>((src_).p1 + 16 * ((S0_) + s_));          \
            r2[s_] = *reinterpret_cast<const float4 *>((src_).p2 + 16 * ((S0_) + s_));          \
            r3[s_] = *reinterpret_cast<const float4 *>((src_).p3 + 16 * ((S0_) + s_));          \
        }                                                                                        \
    }
#define FFN_COMBINE(src_, S0_, dst_)                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < SH; ++s_) {                                          \
        if (tabbed) {                                                                            \
            dst_[(S0_) + s_][0] = ((r1[s_].x * (src_).w1 + r2[s_].x * (src_).w2) + r3[s_].x * (src_).w3) + rx[s_].x * (src_).wx; \
            dst_[(S0_) + s_][1] = ((r1[s_].y * (src_).w1 + r2[s_].y * (src_).w2) + r3[s_].y * (src_).w3) + rx[s_].y * (src_).wx; \
            dst_[(S0_) + s_][2] = ((r1[s_].z * (src_).w1 + r2[s_].z * (src_).w2) + r3[s_].z * (src_).w3) + rx[s_].z * (src_).wx; \
            dst_[(S0_) + s_][3] = ((r1[s_].w * (src_).w1 + r2[s_].w * (src_).w2) + r3[s_].w * (src_).w3) + rx[s_].w * (src_).wx; \
        } else {                                                                                 \
            dst_[(S0_) + s_] = f32x4{rx[s_].x * (src_).wx, rx[s_].y * (src_).wx, rx[s_].z * (src_).wx, rx[s_].w * (src_).wx}; \
        }                                                                                        \
    }

template <int C, int FF>
__global__ void __launch_bounds__(FFS_NW *MSSVT_WAVE) k_ffn_up(FfnArgs a, float *hidden) {
    constexpr int NT = C / 16, HT = FF / 16, LS = C + 4, HG = 4, NG = HT / HG;
    constexpr int SH = NT > 1 ? NT / 2 : 1, NHALF = NT / SH;  // the row is gathered in NHALF pieces
    static_assert(HT % HG == 0, "hidden tiles are walked in groups of 4");
    extern __shared__ float4 lds4[];
    float *W1_l = reinterpret_cast<float *>(lds4);  // [FF][LS]
    float *b1_l = W1_l + FF * LS, *lnw_l = b1_l + FF, *lnb_l = lnw_l + C;
    const int lane = lane_id(), la = lane & 15, g = lane >> 4, wv = threadIdx.x / MSSVT_WAVE;
    const int n = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
    const int tiles = (n + 15) >> 4;
    const bool tabbed = a.tab_row != nullptr;
    if (n <= 0) return;
    int si = 0;
    (void)si;
    STAMP(0)
    // tile k * gridDim + block -> wave k % NW: the last partial round is spread over all CUs
    int tile = wv * gridDim.x + blockIdx.x;
    const bool any = tile < tiles;
    // the first tile's table entry travels while the weights are staged
    FfnTab tab_cur = ffn_tab_load(a, any ? tile : 0, n, la);
    float4 rx[SH], r1[SH], r2[SH], r3[SH];
    f32x4 x[NT], xnext[NT];
    // the first half of the first tile's gather is issued BEFORE the weights are staged: both are pure
    // latency (table -> rows, L2 -> LDS) and overlap
    FfnRowSrc cur = ffn_row_src<C, FF>(a, hidden, any ? tile : 0, n, la, g, tab_cur);
    FFN_ISSUE(cur, 0)
    ffn_stage_weights<C, LS>(W1_l, a.W1, FF);
    for (int e = threadIdx.x; e < FF; e += blockDim.x) b1_l[e] = a.b1[e];
    for (int e = threadIdx.x; e < C; e += blockDim.x) {
        lnw_l[e] = a.ln_w[e];
        lnb_l[e] = a.ln_b[e];
    }
    STAMP(0)
    __syncthreads();
    STAMP(0)
    if (!any) return;
    FfnTab tab_next = ffn_tab_load(a, min(tile + FFS_NW * (int)gridDim.x, tiles - 1), n, la);
#pragma unroll
    for (int hf = 0; hf < NHALF; ++hf) {
        if (hf > 0) FFN_ISSUE(cur, hf * SH)
        FFN_COMBINE(cur, hf * SH, x)
    }
    STAMP(0)
    for (;;) {
        const int tile_next = tile + FFS_NW * gridDim.x;
        const bool has_next = tile_next < tiles;
        // the next tile's gathers are issued below and fly under this tile's MFMAs; its table entry
        // was loaded one tile ago, the one after next is requested now
        FfnRowSrc nxt = cur;
        if (has_next) {
            nxt = ffn_row_src<C, FF>(a, hidden, tile_next, n, la, g, tab_next);
            tab_next = ffn_tab_load(a, min(tile_next + FFS_NW * (int)gridDim.x, tiles - 1), n, la);
        }
        // loads before stores: vmcnt retires in order, a wait for a load also waits for every store
        // issued before it
        if (has_next) FFN_ISSUE(nxt, 0)
        if (cur.live) {  // park x in y: the second launch adds it back
#pragma unroll
            for (int S = 0; S < NT; ++S) *reinterpret_cast<float4 *>(cur.py + 16 * S) = make_float4(x[S][0], x[S][1], x[S][2], x[S][3]);
        }
        // ---- LayerNorm (norm2): a row = the 4 lanes la, la+16, la+32, la+48 --------------------
        float sum = 0.f;
#pragma unroll
        for (int S = 0; S < NT; ++S) sum += (x[S][0] + x[S][1]) + (x[S][2] + x[S][3]);
        sum += lane_xor16(sum);
        sum += lane_xor32(sum);
        const float mean = sum * (1.0f / C);
        float var = 0.f;
#pragma unroll
        for (int S = 0; S < NT; ++S)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = x[S][j] - mean;
                var = __builtin_fmaf(d, d, var);
            }
        var += lane_xor16(var);
        var += lane_xor32(var);
        const float rstd = rsqrtf(var * (1.0f / C) + a.eps);
#pragma unroll
        for (int S = 0; S < NT; ++S) {
            const float4 gw = *reinterpret_cast<const float4 *>(lnw_l + 16 * S + 4 * g);
            const float4 gb = *reinterpret_cast<const float4 *>(lnb_l + 16 * S + 4 * g);
            x[S][0] = (x[S][0] - mean) * rstd * gw.x + gb.x;
            x[S][1] = (x[S][1] - mean) * rstd * gw.y + gb.y;
            x[S][2] = (x[S][2] - mean) * rstd * gw.z + gb.z;
            x[S][3] = (x[S][3] - mean) * rstd * gw.w + gb.w;
        }
        STAMP(0)
        // ---- u^T[hidden][row] = relu(W1 xn + b1), 4 hidden tiles (= 4 independent chains) at a time;
        // the next tile's row gathers are issued / consumed between the groups, i.e. they are in
        // flight under this tile's MFMAs (the branches also pin that order for the scheduler)
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (has_next) {
                if (gi == (NG > 1 ? 1 : 0)) {
                    FFN_COMBINE(nxt, 0, xnext)
                    if (NHALF > 1) FFN_ISSUE(nxt, SH)
                }
                if (NHALF > 1 && gi == NG - 1) FFN_COMBINE(nxt, SH, xnext)
            }
            const int h0 = gi * HG;
            f32x4 acc[HG];
#pragma unroll
            for (int hg = 0; hg < HG; ++hg) {
                const float4 b = *reinterpret_cast<const float4 *>(b1_l + 16 * (h0 + hg) + 4 * g);
                acc[hg] = f32x4{b.x, b.y, b.z, b.w};
            }
            const float *wbase = W1_l + (size_t)(16 * h0 + la) * LS + 4 * g;
            // weights of step S + 1 are read while step S multiplies; the barriers keep the scheduler
            // from hoisting every ds_read of the unrolled loop to the top (VGPRs)
            float4 w[HG], wn[HG];
#pragma unroll
            for (int hg = 0; hg < HG; ++hg) w[hg] = *reinterpret_cast<const float4 *>(wbase + hg * 16 * LS);
#pragma unroll
            for (int S = 0; S < NT; ++S) {
                if (S + 1 < NT) {
#pragma unroll
                    for (int hg = 0; hg < HG; ++hg) wn[hg] = *reinterpret_cast<const float4 *>(wbase + hg * 16 * LS + 16 * (S + 1));
                }
#pragma unroll
                for (int hg = 0; hg < HG; ++hg) MFMA4(acc[hg], w[hg].x, x[S][0]);
#pragma unroll
                for (int hg = 0; hg < HG; ++hg) MFMA4(acc[hg], w[hg].y, x[S][1]);
#pragma unroll
                for (int hg = 0; hg < HG; ++hg) MFMA4(acc[hg], w[hg].z, x[S][2]);
#pragma unroll
                for (int hg = 0; hg < HG; ++hg) MFMA4(acc[hg], w[hg].w, x[S][3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int hg = 0; hg < HG; ++hg) w[hg] = wn[hg];
            }
            if (cur.live) {
#pragma unroll
                for (int hg = 0; hg < HG; ++hg)
                    *reinterpret_cast<float4 *>(cur.pu + 16 * (h0 + hg)) =
                        make_float4(fmaxf(acc[hg][0], 0.f), fmaxf(acc[hg][1], 0.f), fmaxf(acc[hg][2], 0.f),
                                    fmaxf(acc[hg][3], 0.f));
            }
            }
        if (!has_next) break;
#pragma unroll
        for (int S = 0; S < NT; ++S) x[S] = xnext[S];
        cur = nxt;
        tile = tile_next;
    }
}

template <int C, int FF>
__global__ void __launch_bounds__(FFS_NW *MSSVT_WAVE) k_ffn_down(FfnArgs a, const float *hidden) {
    constexpr int NT = C / 16, HT = FF / 16, LS = FF + 4, SG = 4, NG = HT / SG;
    static_assert(HT % SG == 0, "k tiles are walked in groups of 4");
    extern __shared__ float4 lds4[];
    float *W2_l = reinterpret_cast<float *>(lds4);  // [C][LS]
    float *b2_l = W2_l + C * LS, *lnw_l = b2_l + C, *lnb_l = lnw_l + C;
    const int lane = lane_id(), la = lane & 15, g = lane >> 4, wv = threadIdx.x / MSSVT_WAVE;
    const int n = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
    const int tiles = (n + 15) >> 4;
    if (n <= 0) return;
    int tile = wv * gridDim.x + blockIdx.x;
    const bool any = tile < tiles;
    int row = min((any ? tile : 0) * 16 + la, n - 1);
    const float *pu = hidden + (size_t)row * FF + 4 * g;
    // hidden row in groups of 4 k-tiles; the next group's (at the end: the next tile's first group's)
    // loads are in flight during the MFMAs -- and the very first group's while the weights are staged
    float4 ub[SG], un[SG];
#pragma unroll
    for (int s = 0; s < SG; ++s) ub[s] = *reinterpret_cast<const float4 *>(pu + 16 * s);
    ffn_stage_weights<FF, LS>(W2_l, a.W2, C);
    for (int e = threadIdx.x; e < C; e += blockDim.x) {
        b2_l[e] = a.b2[e];
        lnw_l[e] = a.y_norm ? a.ln2_w[e] : 0.f;
        lnb_l[e] = a.y_norm ? a.ln2_b[e] : 0.f;
    }
    __syncthreads();
    if (!any) return;
    for (;;) {
        const int tile_next = tile + FFS_NW * gridDim.x;
        const bool has_next = tile_next < tiles;
        const int row_next = min(tile_next * 16 + la, n - 1);
        const float *pu_next = hidden + (size_t)(has_next ? row_next : row) * FF + 4 * g;
        const bool live = tile * 16 + la < n;
        float *py = a.y + (size_t)row * C + 4 * g;
        f32x4 acc[NT];
        float4 xv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 b = *reinterpret_cast<const float4 *>(b2_l + 16 * t + 4 * g);
            acc[t] = f32x4{b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int S0 = gi * SG;
            if (gi + 1 < NG) {
#pragma unroll
                for (int s = 0; s < SG; ++s) un[s] = *reinterpret_cast<const float4 *>(pu + 16 * (S0 + SG + s));
            } else {
#pragma unroll
                for (int s = 0; s < SG; ++s) un[s] = *reinterpret_cast<const float4 *>(pu_next + 16 * s);
                // x was parked in y by the first launch in this very layout
#pragma unroll
                for (int t = 0; t < NT; ++t) xv[t] = *reinterpret_cast<const float4 *>(py + 16 * t);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float *wbase = W2_l + (size_t)la * LS + 16 * S0 + 4 * g;
#pragma unroll
            for (int s = 0; s < SG; ++s) {
                constexpr int TG = NT > 4 ? 4 : NT;  // output tiles per group: 4 independent chains
#pragma unroll
                for (int t0 = 0; t0 < NT; t0 += TG) {
                    float4 w[TG];
#pragma unroll
                    for (int t = 0; t < TG; ++t) w[t] = *reinterpret_cast<const float4 *>(wbase + (t0 + t) * 16 * LS + 16 * s);
#pragma unroll
                    for (int t = 0; t < TG; ++t) MFMA4(acc[t0 + t], w[t].x, ub[s].x);
#pragma unroll
                    for (int t = 0; t < TG; ++t) MFMA4(acc[t0 + t], w[t].y, ub[s].y);
#pragma unroll
                    for (int t = 0; t < TG; ++t) MFMA4(acc[t0 + t], w[t].z, ub[s].z);
#pragma unroll
                    for (int t = 0; t < TG; ++t) MFMA4(acc[t0 + t], w[t].w, ub[s].w);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SG; ++s) ub[s] = un[s];
        }
        // ---- y = x + (W2 u + b2) -----------------------------------------------------------------
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t][0] += xv[t].x; acc[t][1] += xv[t].y; acc[t][2] += xv[t].z; acc[t][3] += xv[t].w;
            sum += (acc[t][0] + acc[t][1]) + (acc[t][2] + acc[t][3]);
        }
        if (live) {
#pragma unroll
            for (int t = 0; t < NT; ++t) *reinterpret_cast<float4 *>(py + 16 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        }
        if (a.y_norm) {  // LayerNorm of y for the next block
            sum += lane_xor16(sum);
            sum += lane_xor32(sum);
            const float mean = sum * (1.0f / C);
            float var = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = acc[t][j] - mean;
                    var = __builtin_fmaf(d, d, var);
                }
            var += lane_xor16(var);
            var += lane_xor32(var);
            const float rstd = rsqrtf(var * (1.0f / C) + a.eps2);
            if (live) {
                float *pn = a.y_norm + (size_t)row * C + 4 * g;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 gw = *reinterpret_cast<const float4 *>(lnw_l + 16 * t + 4 * g);
                    const float4 gb = *reinterpret_cast<const float4 *>(lnb_l + 16 * t + 4 * g);
                    *reinterpret_cast<float4 *>(pn + 16 * t) =
                        make_float4((acc[t][0] - mean) * rstd * gw.x + gb.x, (acc[t][1] - mean) * rstd * gw.y + gb.y,
                                    (acc[t][2] - mean) * rstd * gw.z + gb.z, (acc[t][3] - mean) * rstd * gw.w + gb.w);
                }
            }
        }
        if (!has_next) break;
        tile = tile_next;
        row = row_next;
        pu = pu_next;
    }
}

// =====================================================================================================
// Single-launch FFN tail, weights stationary in registers, fp32 operands split into two fp16 halves
// =====================================================================================================
// The fp32 matrix instruction above runs at 1/16 of the 16-bit rate, and at 62 us per FFN (both launches) it bounds
// the frame.  Here every fp32 operand v is split into  hi = fp16(v)  and  lo = fp16((v - hi) * 2^11)  (round toward zero: 22+
// mantissa bits together; the lo half is scaled so that it stays out of the fp16 subnormals) and a product sum is three
// v_mfma_f32_16x16x32_f16 with fp32 accumulation:
//        sum a b  =  sum a_hi b_hi  +  2^-11 (sum a_hi b_lo + sum a_lo b_hi)            (a_lo b_lo ~ 2^-22: dropped)
// 3/16 of the fp32 instruction's cycles for the same result to ~2^-21 relative per operand (measured against float64:
// the same max error as the fp32 kernels).  The caller checks the fp16 RANGE of the operands from
// the parameters (|LayerNorm output| <= sqrt(C) max|w| + max|b|, |hidden| <= max_h(|W1_h|_1 xmax + |b1_h|) < 6e4 = fused.FFN_F16_LIMIT) and
// keeps the fp32 kernels otherwise.  (Not exact: hi + 2^-11 lo drops the last ~2 mantissa bits of v and the product sum drops
// a_lo b_lo; both ~2^-22 relative.)
//
// With the matrix work that cheap the layout changes: one workgroup of FF/32 = C/16 waves per CU; wave w keeps, as
// MFMA A-fragments in REGISTERS for the whole launch (2 x 64 VGPRs at C = 128), the W1 rows of hidden units
// [32 w, 32 w + 32) and the W2 rows of output channels [16 w, 16 w + 16) -- no weight traffic through LDS, none per
// tile.  A tile of 16 rows is
//   A. loaded row-wise by the whole workgroup (a lane = 4 channels of a row: coalesced 512-byte rows, the residual
//      input built on the fly, LayerNorm by a DPP reduction over the row's lanes), split and written to LDS as the
//      B-fragments of GEMM1 (8 KB);                                                                     [barrier]
//   B. multiplied by every wave with ITS W1 slice: u^T[32 hidden][16 rows].  The accumulator layout of the 16x16 MFMA
//      is the B-operand layout of the next one (k slot (g, j) <-> hidden 16 (j / 4) + 4 g + j % 4), so u is split in
//      place and published as this wave's k-slice of GEMM2's B-fragments (2 KB per wave);               [barrier]
//   C. every wave multiplies ALL k-slices by its W2 rows: 16 output channels x 16 rows, complete sums in a fixed
//      order (deterministic), + b2, into a 16 x C tile in LDS;                                          [barrier]
//   D. the lane that still holds x for a row piece adds it, applies the next block's LayerNorm and stores whole rows.
// The (N, FF) hidden activations never exist in memory and x is read once: 131 MB of HBM traffic per FFN at 74k rows
// instead of 270 MB.  The phases of consecutive tiles are software-pipelined (see the loop) so that each barrier
// interval has one MFMA phase and one row-wise phase: two barriers per tile.  (First version: split-K over
// the hidden units in GEMM2 with the FF/32 partial tiles summed through LDS -- 128 more VALU instructions per wave
// and tile for the partial epilogues than the third barrier costs; the phases are VALU-issue bound, not MFMA bound.)
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
#define MFMA_H(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16((av), (bv), acc, 0, 0, 0)
#define FFW_SCALE 2048.0f
#define FFW_INV (1.0f / 2048.0f)

// Row-wise arithmetic on PAIRS: the phases between the matrix products are VALU-issue bound (~350 vector instructions per
// wave and 16-row tile against 48 MFMAs), and v_pk_mul / v_pk_add / v_pk_fma_f32 do two lanes' worth of the same IEEE
// operation per issue slot.  Same operations in the same order per element: results are bit-identical to the scalar form.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk2(float a, float b) { return f32x2{a, b}; }
__device__ __forceinline__ f32x2 pk1(float a) { return f32x2{a, a}; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 ffw_relu2(f32x2 v) { return f32x2{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)}; }

// v -> (hi, lo) halves of 4 floats
__device__ __forceinline__ void ffw_split4(const f32x2 v01, const f32x2 v23, h16x4 &hi, h16x4 &lo) {
    const fp16x2 a = __builtin_amdgcn_cvt_pkrtz(v01[0], v01[1]), b = __builtin_amdgcn_cvt_pkrtz(v23[0], v23[1]);
    // (v - hi) 2^11 as fma(hi, -2^11, v 2^11): exact steps, the bits of the subtraction form; the fp16 -> fp32 conversion
    // of hi rides inside v_fma_mix_f32 (one packed multiply + two mixed fmas per pair instead of two conversions + two packed ops)
    const f32x2 s01 = v01 * pk1(FFW_SCALE), s23 = v23 * pk1(FFW_SCALE);
    const fp16x2 c = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)a[0], -FFW_SCALE, s01[0]), __builtin_fmaf((float)a[1], -FFW_SCALE, s01[1])),
                 d = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)b[0], -FFW_SCALE, s23[0]), __builtin_fmaf((float)b[1], -FFW_SCALE, s23[1]));
    hi = h16x4{(_Float16)a[0], (_Float16)a[1], (_Float16)b[0], (_Float16)b[1]};
    lo = h16x4{(_Float16)c[0], (_Float16)c[1], (_Float16)d[0], (_Float16)d[1]};
}
__device__ __forceinline__ void ffw_split4(const float v0, const float v1, const float v2, const float v3, h16x4 &hi, h16x4 &lo) {
    ffw_split4(pk2(v0, v1), pk2(v2, v3), hi, lo);
}
__device__ __forceinline__ h16x8 ffw_cat(const h16x4 a, const h16x4 b) {
    return h16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ void ffw_split8(const float *p0, const float *p1, h16x8 &hi, h16x8 &lo) {
    const float4 v0 = *reinterpret_cast<const float4 *>(p0), v1 = *reinterpret_cast<const float4 *>(p1);
    h16x4 h0, l0, h1, l1;
    ffw_split4(v0.x, v0.y, v0.z, v0.w, h0, l0);
    ffw_split4(v1.x, v1.y, v1.z, v1.w, h1, l1);
    hi = ffw_cat(h0, h1);
    lo = ffw_cat(l0, l1);
}

template <int L>
__device__ __forceinline__ float ffw_row_sum(float v) {  // all-reduce over L consecutive lanes (L = 8, 16, 32)
    v += DPP_MOV(v, 0xB1);
    v += DPP_MOV(v, 0x4E);
    v += DPP_MOV(v, 0x141);
    if (L >= 16) v += DPP_MOV(v, 0x140);
    if (L >= 32) v += lane_xor16(v);
    return v;
}

// A-fragments of wave wv, lane (la, g): W1 rows 32 wv + 16 T + la, k slot (g, j) <-> channel 32 P + 8 g + j;
// W2 rows 16 wv + la, k slice ks: k slot (g, j) <-> hidden 32 ks + 16 (j / 4) + 4 g + j % 4 (the accumulator layout of GEMM1)
template <int C, int FF>
__device__ __forceinline__ void ffw_weight_frags(const float *W1, const float *W2, int wv, int la, int g,
                                                 h16x8 (&W1h)[2][C / 32], h16x8 (&W1l)[2][C / 32], h16x8 (&W2h)[FF / 32],
                                                 h16x8 (&W2l)[FF / 32]) {
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int P = 0; P < C / 32; ++P) {
            const float *src = W1 + (size_t)(32 * wv + 16 * T + la) * C + 32 * P + 8 * g;
            ffw_split8(src, src + 4, W1h[T][P], W1l[T][P]);
        }
#pragma unroll
    for (int ks = 0; ks < FF / 32; ++ks) {
        const float *src = W2 + (size_t)(16 * wv + la) * FF + 32 * ks + 4 * g;
        ffw_split8(src, src + 16, W2h[ks], W2l[ks]);
    }
}

// the fragments of all waves, split once per parameter version: [wave][fragment][lane] x 16 bytes
template <int C, int FF>
__global__ void __launch_bounds__(MSSVT_WAVE) k_ffn_pack(const float *W1, const float *W2, h16x8 *packed) {
    constexpr int NP = C / 32, NW = FF / 32, NF = 4 * NP + 2 * NW;
    const int wv = blockIdx.x, lane = lane_id();
    h16x8 W1h[2][NP], W1l[2][NP], W2h[NW], W2l[NW];
    ffw_weight_frags<C, FF>(W1, W2, wv, lane & 15, lane >> 4, W1h, W1l, W2h, W2l);
    h16x8 *dst = packed + (size_t)wv * NF * 64 + lane;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int P = 0; P < NP; ++P) {
            dst[((T * NP + P) * 2) * 64] = W1h[T][P];
            dst[((T * NP + P) * 2 + 1) * 64] = W1l[T][P];
        }
#pragma unroll
    for (int ks = 0; ks < NW; ++ks) {
        dst[(4 * NP + 2 * ks) * 64] = W2h[ks];
        dst[(4 * NP + 2 * ks + 1) * 64] = W2l[ks];
    }
}

#ifdef MSSVT_STAMPS
__device__ unsigned long long g_ws_stamps[4 * 8 * 16];  // [block < 4][wave][phase]
extern "C" int mssvt_debug_read_ffn_ws_stamps(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws_stamps), sizeof(g_ws_stamps));
}
// start / end of every workgroup on the 100 MHz wall clock (comparable across CUs): launch skew and tail of the grid
__device__ unsigned long long g_ws_span[1024 * 2];
extern "C" int mssvt_debug_read_ffn_ws_spans(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws_span), sizeof(g_ws_span));
}
#define WSTAMP(k_) { const unsigned long long t_ = __builtin_readcyclecounter(); ws_acc[k_] += t_ - ws_t; ws_t = t_; }
#else
#define WSTAMP(k_)
#endif

// STOREY = false (round 6): y itself has no reader -- the last Block in front of a CompressBlock, which takes only the
// LayerNorm output (it has no input residual, ref mssvt_backbone.py:370-385) -- and is not stored: 4 C of the launch's 12 C
// bytes per row
template <int C, int FF, bool TABBED, bool NORM2, bool STOREY = true>
__global__ void __launch_bounds__((FF / 32) * MSSVT_WAVE, 1) k_ffn_ws(FfnArgs a, const h16x8 *packed) {
#ifdef MSSVT_STAMPS
    unsigned long long ws_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ws_t = __builtin_readcyclecounter();
    int ws_tiles = 0;
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_ws_span[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int NW = FF / 32, NP = C / 32, LPR = C / 4, RPW = MSSVT_WAVE / LPR, PS = C + 4;
    static_assert(FF == 2 * C && NW * RPW == 16 && NW * 16 == C && (LPR == 8 || LPR == 16 || LPR == 32),
                  "16 rows per workgroup tile, one 16-channel output tile per wave");
    extern __shared__ float4 lds4[];
    h16x8 *bfrag = reinterpret_cast<h16x8 *>(lds4);  // [NP][hi | lo][64 slots]   B operands of GEMM1
    h16x8 *ufrag = bfrag + NP * 2 * 64;              // [NW][hi | lo][64 lanes]   B operands of GEMM2, one k-slice per wave
    float *ytile = reinterpret_cast<float *>(ufrag + NW * 2 * 64);  // [16 rows][PS]
    float *lnp = ytile + 16 * PS;                                   // [norm2 w | norm2 b | next norm w | next norm b][C]
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    const int r = wv * RPW + lane / LPR, q = lane % LPR;  // row-wise view: row of the tile, channels [4 q, 4 q + 4)
    const int n = a.n_rows_dev ? *a.n_rows_dev : a.n_rows;
    const int tiles = (n + 15) >> 4;
    // Tiles are dealt so that an XCD owns ONE contiguous eighth of the rows and its gridDim / 8 workgroups walk it side by side
    // (workgroups are dispatched round-robin over the XCDs: block b runs on XCD b % 8 -- observed, used for speed only): the
    // voxels of a window lie in three x-slabs ~160 rows apart in the (b, x, y, z) order and gather the SAME attention rows, so
    // the rows a workgroup gathers were mostly fetched into ITS XCD's L2 by a neighbour a moment ago.  Round-robin tiles put
    // consecutive tiles on different XCDs (each with an L2 of its own): 146 MB of HBM traffic per launch against 114 MB
    // algorithmic.  MSSVT_XCD_REMAP=0 restores the round-robin deal.
    int tstep = gridDim.x, tend = tiles, tile0 = blockIdx.x;
    if (a.xcd && (gridDim.x & 7) == 0 && tiles >= 8 * (int)gridDim.x) {
        const int x = blockIdx.x & 7, per_xcd = (tiles + 7) >> 3;
        tstep = gridDim.x >> 3;
        tile0 = x * per_xcd + (int)(blockIdx.x >> 3);
        tend = min((x + 1) * per_xcd, tiles);
    }
    if (n <= 0 || tile0 >= tend) return;

    // Rows past the end are CLAMPED to row n - 1 everywhere (table, gathers, stores): such lanes compute exactly what the
    // lanes of row n - 1 compute and store the same values to the same place -- no predication, hence no branches
    // inside the barrier intervals (the scheduler interleaves MFMA and VALU only within one basic block).
    int tile = tile0;
    int4 tr = make_int4(0, 0, 0, 0);
    float4 tw = make_float4(0.f, 0.f, 0.f, 0.f);
    int own = 0;
#define FFW_TAB(tile_, tr_, tw_, own_)                                                                      \
    {                                                                                                       \
        const int row_ = min((tile_) * 16 + r, n - 1);                                                      \
        if (TABBED) { tr_ = a.tab_row[row_]; tw_ = a.tab_w[row_]; }                                         \
        else if (a.owner) own_ = a.owner[row_];                                                             \
    }
    float4 rx, r1, r2, r3;
    float w1 = 0.f, w2 = 0.f, w3 = 0.f, wx = 1.f;
#define FFW_SELECT(tr_, tw_, row_)                                                                          \
            /* unowned voxel (tr.x < 0): 2 x_in; it re-reads its own finite row with weight 0.  The selection is   \
               written with MASKS, not ?: -- the compiler turns a group of selects on one condition into a BRANCH,   \
               i.e. a basic-block boundary inside the barrier interval (MFMA and VALU interleave only within a      \
               block), and with ?: on the pointers it even replaced one gather by "copy rx", which waits for the    \
               row load issued a few instructions earlier */                                                        \
            const int m_ = ~(tr_.x >> 31); /* owned: ~0, unowned: 0 */                                              \
            const long long dm_ = ((const char *)a.x_in - (const char *)a.attn) & (long long)~m_;                   \
            const char *b_ = (const char *)a.attn + dm_ + 16 * q;                                                   \
            const float *px_ = a.x_in + (size_t)row_ * C + 4 * q;                                                   \
            const int un_row_ = row_ & ~m_;                                                                         \
            rx = *reinterpret_cast<const float4 *>(px_);                                                            \
            r1 = *reinterpret_cast<const float4 *>(b_ + (size_t)((tr_.x & m_) | un_row_) * (C * 4));               \
            r2 = *reinterpret_cast<const float4 *>(b_ + (size_t)((tr_.y & m_) | un_row_) * (C * 4));               \
            r3 = *reinterpret_cast<const float4 *>(b_ + (size_t)((tr_.z & m_) | un_row_) * (C * 4));               \
            w1 = __builtin_bit_cast(float, __builtin_bit_cast(int, tw_.x) & m_);                                    \
            w2 = __builtin_bit_cast(float, __builtin_bit_cast(int, tw_.y) & m_);                                    \
            w3 = __builtin_bit_cast(float, __builtin_bit_cast(int, tw_.z) & m_);                                    \
            wx = __builtin_bit_cast(float, 0x3F800000 + (~m_ & 0x00800000)); /* 1.0f, or 2.0f when unowned */
#define FFW_ISSUE(tile_, tr_, tw_, own_)                                                                    \
    {                                                                                                       \
        const int row_ = min((tile_) * 16 + r, n - 1);                                                      \
        if (TABBED) {                                                                                       \
            FFW_SELECT(tr_, tw_, row_)                                                                      \
        } else {                                                                                            \
            const bool dbl_ = a.owner != nullptr && own_ < 0;                                               \
            rx = *reinterpret_cast<const float4 *>((dbl_ ? a.x_in : a.x_new) + (size_t)row_ * C + 4 * q);   \
            wx = dbl_ ? 2.0f : 1.0f;                                                                        \
        }                                                                                                   \
    }
#define FFW_COMBINE(x_)                                                                                     \
    if (TABBED) {                                                                                           \
        const f32x2 a_ = ((pk2(r1.x, r1.y) * pk1(w1) + pk2(r2.x, r2.y) * pk1(w2)) + pk2(r3.x, r3.y) * pk1(w3)) + pk2(rx.x, rx.y) * pk1(wx); \
        const f32x2 b_ = ((pk2(r1.z, r1.w) * pk1(w1) + pk2(r2.z, r2.w) * pk1(w2)) + pk2(r3.z, r3.w) * pk1(w3)) + pk2(rx.z, rx.w) * pk1(wx); \
        x_ = make_float4(a_[0], a_[1], b_[0], b_[1]);                                                       \
    } else {                                                                                                \
        const f32x2 a_ = pk2(rx.x, rx.y) * pk1(wx), b_ = pk2(rx.z, rx.w) * pk1(wx);                          \
        x_ = make_float4(a_[0], a_[1], b_[0], b_[1]);                                                       \
    }
    // LayerNorm (norm2) over the row's LPR lanes, split, B fragments of GEMM1.  channel 4 q = 32 P + 8 gq + j0: fragment
    // P, lane (la = r, g = gq), halves j0 .. j0 + 3.  Slot of lane (la, g) inside a fragment: 16 g + ((la + rot(g) + P) & 15)
    // with rot = 0, 0, 4, 4.  ds_read_b128 serves a wave in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32):
    // a group takes la in {0..3, 12..15} from g and la in {4..11} from g + 1 -- complementary sets, so g and g + 1 must carry the
    // SAME rotation for its 16 slots to differ mod 16 (rounds 3 - 5 rotated by 4 g, which is conflict free for 16 CONTIGUOUS
    // lanes but made lanes 12-15 and 24-27 of every read collide: 22 % of the kernel's LDS cycles were bank conflicts); the
    // writers of a row (16 (P, gq) lanes, 8 bytes each) stay at the two-way level of the old form through the 4-slot step
    // between the g pairs (plain 16 g + la: four-way)
#define FFW_ROT(g_) (((g_) >> 1) << 2)
#define FFW_NORM_TO_BFRAG(x_)                                                                               \
    {                                                                                                       \
        const float mean_ = ffw_row_sum<LPR>((x_.x + x_.y) + (x_.z + x_.w)) * (1.0f / C);                   \
        const f32x2 d01_ = pk2(x_.x, x_.y) - pk1(mean_), d23_ = pk2(x_.z, x_.w) - pk1(mean_);               \
        const float var_ = ffw_row_sum<LPR>(                                                                \
            __builtin_fmaf(d23_[1], d23_[1], __builtin_fmaf(d23_[0], d23_[0], __builtin_fmaf(d01_[1], d01_[1], d01_[0] * d01_[0])))); \
        const float rstd_ = rsqrtf(var_ * (1.0f / C) + a.eps);                                              \
        h16x4 hi_, lo_;                                                                                     \
        const float4 lnw = FFW_LNW, lnb = FFW_LNB;                                                          \
        ffw_split4(d01_ * pk1(rstd_) * pk2(lnw.x, lnw.y) + pk2(lnb.x, lnb.y),                               \
                   d23_ * pk1(rstd_) * pk2(lnw.z, lnw.w) + pk2(lnb.z, lnb.w), hi_, lo_);                     \
        const int P_ = q >> 3, gq_ = (q >> 1) & 3, j0_ = (q & 1) * 4;                                       \
        h16x4 *dst_ = reinterpret_cast<h16x4 *>(bfrag + (P_ * 2) * 64 + 16 * gq_ + ((r + FFW_ROT(gq_) + P_) & 15)) + (j0_ >> 2); \
        dst_[0] = hi_;                                                                                      \
        dst_[64 * 2] = lo_; /* the lo fragment follows the hi fragment: 64 slots x 2 h16x4 */               \
    }
#define FFW_GEMM1_PRODUCTS()                                                                                \
        _Pragma("unroll") for (int P = 0; P < NP; ++P) {                                                   \
            _Pragma("unroll") for (int T = 0; T < 2; ++T) MFMA_H(um_[T], W1h[T][P], bh_[P]);               \
            _Pragma("unroll") for (int T = 0; T < 2; ++T) MFMA_H(ul_[T], W1h[T][P], bl_[P]);               \
            _Pragma("unroll") for (int T = 0; T < 2; ++T) MFMA_H(uk_[T], W1l[T][P], bh_[P]);               \
        }
    // u^T = relu(W1 xn + b1) for this wave's 32 hidden units -> its k-slice of GEMM2's B operand
#define FFW_GEMM1()                                                                                         \
    {                                                                                                       \
        f32x4 um_[2], ul_[2], uk_[2]; /* hi hi | hi lo | lo hi: six independent accumulation chains */      \
        h16x8 bh_[NP], bl_[NP];                                                                             \
        _Pragma("unroll") for (int P = 0; P < NP; ++P) {                                                   \
            const int slot_ = 16 * g + ((la + FFW_ROT(g) + P) & 15);                                             \
            bh_[P] = bfrag[(P * 2) * 64 + slot_];                                                           \
            bl_[P] = bfrag[(P * 2 + 1) * 64 + slot_];                                                       \
        }                                                                                                   \
        _Pragma("unroll") for (int T = 0; T < 2; ++T) {                                                    \
            um_[T] = f32x4{bias1[T][0], bias1[T][1], bias1[T][2], bias1[T][3]};                             \
            ul_[T] = f32x4{0.f, 0.f, 0.f, 0.f};                                                             \
            uk_[T] = f32x4{0.f, 0.f, 0.f, 0.f};                                                             \
        }                                                                                                   \
        FFW_GEMM1_PRODUCTS()                                                                                \
        h16x4 h0_, l0_, h1_, l1_;                                                                           \
        ffw_split4(FFW_U2(0, 0), FFW_U2(0, 2), h0_, l0_);                                                   \
        ffw_split4(FFW_U2(1, 0), FFW_U2(1, 2), h1_, l1_);                                                   \
        ufrag[(wv * 2) * 64 + lane] = ffw_cat(h0_, h1_);                                                    \
        ufrag[(wv * 2 + 1) * 64 + lane] = ffw_cat(l0_, l1_);                                                \
    }
#define FFW_U(T_, i_) fmaxf(__builtin_fmaf(ul_[T_][i_] + uk_[T_][i_], FFW_INV, um_[T_][i_]), 0.f)
    // elements i_, i_ + 1 at once: relu(fma(ul + uk, 2^-11, um))
#define FFW_U2(T_, i_) ffw_relu2(pk_fma(pk2(ul_[T_][i_], ul_[T_][i_ + 1]) + pk2(uk_[T_][i_], uk_[T_][i_ + 1]), pk1(FFW_INV), \
                                        pk2(um_[T_][i_], um_[T_][i_ + 1])))

    // ---- the first tile's rows are requested before the weights (both pure latency) -------------------------
    FFW_TAB(tile, tr, tw, own)
    FFW_ISSUE(tile, tr, tw, own)
    int4 trn = tr;
    float4 twn = tw;
    int ownn = own;
    FFW_TAB(min(tile + tstep, tend - 1), trn, twn, ownn)

    // ---- this wave's weight slices as A fragments: pre-split by k_ffn_pack, or split here ---------------------
    h16x8 W1h[2][NP], W1l[2][NP], W2h[NW], W2l[NW];
    if (packed) {
        const h16x8 *src = packed + (size_t)wv * (4 * NP + 2 * NW) * 64 + lane;
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int P = 0; P < NP; ++P) {
                W1h[T][P] = src[((T * NP + P) * 2) * 64];
                W1l[T][P] = src[((T * NP + P) * 2 + 1) * 64];
            }
#pragma unroll
        for (int ks = 0; ks < NW; ++ks) {
            W2h[ks] = src[(4 * NP + 2 * ks) * 64];
            W2l[ks] = src[(4 * NP + 2 * ks + 1) * 64];
        }
    } else {
        ffw_weight_frags<C, FF>(a.W1, a.W2, wv, la, g, W1h, W1l, W2h, W2l);
    }
    float bias1[2][4];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int i = 0; i < 4; ++i) bias1[T][i] = a.b1[32 * wv + 16 * T + 4 * g + i];
    const float4 bias2 = *reinterpret_cast<const float4 *>(a.b2 + 16 * wv + 4 * g);  // MFMA view: channels 16 wv + 4 g + i
    // row-wise constants of this lane's 4 channels
    // (kept in LDS, not in 16 VGPRs: with the weights in 128 registers the loop sits at the 256-register limit, and every
    // register less is a spill or a v_mov less; four ds_read_b128 per tile)
    if (threadIdx.x < C / 4) {
        const int c4 = threadIdx.x * 4;
        *reinterpret_cast<float4 *>(lnp + c4) = *reinterpret_cast<const float4 *>(a.ln_w + c4);
        *reinterpret_cast<float4 *>(lnp + C + c4) = *reinterpret_cast<const float4 *>(a.ln_b + c4);
        if (NORM2) {
            *reinterpret_cast<float4 *>(lnp + 2 * C + c4) = *reinterpret_cast<const float4 *>(a.ln2_w + c4);
            *reinterpret_cast<float4 *>(lnp + 3 * C + c4) = *reinterpret_cast<const float4 *>(a.ln2_b + c4);
        }
    }
    __syncthreads();
#define FFW_LNW *reinterpret_cast<const float4 *>(lnp + 4 * q)
#define FFW_LNB *reinterpret_cast<const float4 *>(lnp + C + 4 * q)
#define FFW_LN2W *reinterpret_cast<const float4 *>(lnp + 2 * C + 4 * q)
#define FFW_LN2B *reinterpret_cast<const float4 *>(lnp + 3 * C + 4 * q)

    // Software pipeline over the tiles, two barrier intervals per tile, each holding one MFMA phase and one row-wise
    // (VALU / memory) phase of a DIFFERENT tile so that the matrix pipe and the vector ALU overlap:
    //   I2(t) = { GEMM2(t) -> y tile  |  A(t+1): x, LayerNorm, split -> B fragments; row gathers of t+2 leave }  [barrier]
    //   I1(t) = { D(t): y tile + x -> y, next LayerNorm, stores  |  GEMM1(t+1) -> this wave's k-slice of u }     [barrier]
    // bfrag(t+1) is written while the others may still be in GEMM2(t) (everybody is past GEMM1(t)); ufrag(t+1) and the
    // reads of ytile(t) share an interval (everybody is past GEMM2(t)).  Past the last tile the pipeline repeats the
    // last tile (A and GEMM1 once more, results unused) instead of branching.
    float4 xc;
    FFW_COMBINE(xc)
    {
        const int t1 = min(tile + tstep, tend - 1);
        FFW_ISSUE(t1, trn, twn, ownn)
        FFW_TAB(min(t1 + tstep, tend - 1), trn, twn, ownn)
    }
    FFW_NORM_TO_BFRAG(xc)
    __syncthreads();
    FFW_GEMM1()
    WSTAMP(0)
    __syncthreads();
    // The two intervals are written as CHUNKS fenced by sched_barrier(0): a chunk = the three products of one k-step (one
    // hi hi, one hi lo, one lo hi MFMA: independent accumulators) + the LDS reads of the next step + one PIECE of the
    // row-wise work of the other tile.  Left to itself the scheduler either serialises the two phases (every MFMA, then
    // every vector instruction: both waves of a SIMD then want the same pipe at the same time) or hoists every LDS read to
    // the top (spills at 256 registers); sched_group_barrier patterns were tried and are not stable from build to build.
    // the later-dispatched half of the workgroup -- the loser of every issue arbitration on its SIMD, MI355X_MICROARCH.md "Two
    // waves per SIMD" item 4 -- runs at static priority 1: 46.4 -> 46.0 and 47.4 -> 46.5 us on two boxes (MSSVT_FFN_PRIO=0: off)
    if (a.prio && wv >= NW / 2) __builtin_amdgcn_s_setprio(1);
    for (;;) {
        const int tile_next = tile + tstep;
        const bool has_next = tile_next < tend;
        const int t1 = has_next ? tile_next : tile;  // the tile whose A / GEMM1 run in this iteration
        WSTAMP(1)
        // ---- I2: GEMM2(t): this wave's 16 output channels over all k-slices | A(t1) ------------------------------
        float4 xn;
        {
            constexpr int NPC2 = 6, PPS2 = (NPC2 + NW - 1) / NW;  // pieces of A, pieces per k-step
            f32x4 m = f32x4{bias2.x, bias2.y, bias2.z, bias2.w}, l = f32x4{0.f, 0.f, 0.f, 0.f}, k = l;
            h16x8 uh = ufrag[lane], ulo = ufrag[64 + lane];
            f32x2 d01_ = pk1(0.f), d23_ = pk1(0.f), n01_ = pk1(0.f), n23_ = pk1(0.f);
            float mean_ = 0.f, var_ = 0.f;
            (void)mean_;
#pragma unroll
            for (int ks = 0; ks < NW; ++ks) {
                h16x8 uhn = uh, ulon = ulo;
                if (ks + 1 < NW) {
                    uhn = ufrag[((ks + 1) * 2) * 64 + lane];
                    ulon = ufrag[((ks + 1) * 2 + 1) * 64 + lane];
                }
                MFMA_H(m, W2h[ks], uh);
                MFMA_H(l, W2h[ks], ulo);
                MFMA_H(k, W2l[ks], uh);
#pragma unroll
                for (int pc = ks * PPS2; pc < (ks + 1) * PPS2 && pc < NPC2; ++pc) {
                    if (pc == 0) {
                        FFW_COMBINE(xn)
                    } else if (pc == 1) {
                        const int t2 = min(t1 + tstep, tend - 1);
                        FFW_ISSUE(t2, trn, twn, ownn)
                        FFW_TAB(min(t2 + tstep, tend - 1), trn, twn, ownn)
                    } else if (pc == 2) {
                        mean_ = ffw_row_sum<LPR>((xn.x + xn.y) + (xn.z + xn.w)) * (1.0f / C);
                        d01_ = pk2(xn.x, xn.y) - pk1(mean_);
                        d23_ = pk2(xn.z, xn.w) - pk1(mean_);
                    } else if (pc == 3) {
                        var_ = ffw_row_sum<LPR>(__builtin_fmaf(
                            d23_[1], d23_[1], __builtin_fmaf(d23_[0], d23_[0], __builtin_fmaf(d01_[1], d01_[1], d01_[0] * d01_[0]))));
                    } else if (pc == 4) {
                        const float rstd_ = rsqrtf(var_ * (1.0f / C) + a.eps);
                        const float4 lnw = FFW_LNW, lnb = FFW_LNB;
                        n01_ = d01_ * pk1(rstd_) * pk2(lnw.x, lnw.y) + pk2(lnb.x, lnb.y);
                        n23_ = d23_ * pk1(rstd_) * pk2(lnw.z, lnw.w) + pk2(lnb.z, lnb.w);
                    } else {
                        h16x4 hi_, lo_;
                        ffw_split4(n01_, n23_, hi_, lo_);
                        const int P_ = q >> 3, gq_ = (q >> 1) & 3, j0_ = (q & 1) * 4;
                        h16x4 *dst_ = reinterpret_cast<h16x4 *>(bfrag + (P_ * 2) * 64 + 16 * gq_ + ((r + FFW_ROT(gq_) + P_) & 15)) + (j0_ >> 2);
                        dst_[0] = hi_;
                        dst_[64 * 2] = lo_;
                    }
                }
                uh = uhn;
                ulo = ulon;
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                const f32x2 y01 = pk_fma(pk2(l[0], l[1]) + pk2(k[0], k[1]), pk1(FFW_INV), pk2(m[0], m[1])),
                            y23 = pk_fma(pk2(l[2], l[3]) + pk2(k[2], k[3]), pk1(FFW_INV), pk2(m[2], m[3]));
                *reinterpret_cast<float4 *>(ytile + la * PS + 16 * wv + 4 * g) = make_float4(y01[0], y01[1], y23[0], y23[1]);
            }
        }
        WSTAMP(2)
        __syncthreads();
        WSTAMP(3)
        // ---- I1: D(t): y = x + (W2 u + b2), the next block's LayerNorm, whole rows out | GEMM1(t1) ------------------
        {
            constexpr int NS1 = 2 * NP, NPC1 = NORM2 ? 5 : 2;  // steps (T, P); pieces of D + the split of hidden tile 0
            const size_t row = (size_t)min(tile * 16 + r, n - 1);
            f32x4 um_[2], ul_[2], uk_[2];
            h16x8 bh_[NP], bl_[NP];
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                um_[T] = f32x4{bias1[T][0], bias1[T][1], bias1[T][2], bias1[T][3]};
                ul_[T] = f32x4{0.f, 0.f, 0.f, 0.f};
                uk_[T] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            bh_[0] = bfrag[16 * g + ((la + FFW_ROT(g)) & 15)];
            bl_[0] = bfrag[64 + 16 * g + ((la + FFW_ROT(g)) & 15)];
            const float4 yt = *reinterpret_cast<const float4 *>(ytile + r * PS + 4 * q);
            f32x2 y01 = pk1(0.f), y23 = y01, d01 = y01, d23 = y01;
            float var = 0.f;
            h16x4 h0_, l0_, h1_, l1_;
#pragma unroll
            for (int st = 0; st < NS1; ++st) {
                const int T = st / NP, P = st % NP;
                if (T == 0 && P + 1 < NP) {
                    const int slot_ = 16 * g + ((la + FFW_ROT(g) + P + 1) & 15);
                    bh_[P + 1] = bfrag[((P + 1) * 2) * 64 + slot_];
                    bl_[P + 1] = bfrag[((P + 1) * 2 + 1) * 64 + slot_];
                }
                MFMA_H(um_[T], W1h[T][P], bh_[P]);
                MFMA_H(ul_[T], W1h[T][P], bl_[P]);
                MFMA_H(uk_[T], W1l[T][P], bh_[P]);
                // pieces of D spread over the first NP steps; the split of hidden tile 0 one step after its last product
                constexpr int PPS1 = (4 + NP - 1) / NP;
#pragma unroll
                for (int pc = st * PPS1; pc < (st + 1) * PPS1 && pc < 4 && st < NP; ++pc) {
                    if (pc == 0) {
                        y01 = pk2(yt.x, yt.y) + pk2(xc.x, xc.y);
                        y23 = pk2(yt.z, yt.w) + pk2(xc.z, xc.w);
                        if (STOREY) *reinterpret_cast<float4 *>(a.y + row * C + 4 * q) = make_float4(y01[0], y01[1], y23[0], y23[1]);
                    } else if (pc == 1 && NORM2) {
                        const float mean = ffw_row_sum<LPR>((y01[0] + y01[1]) + (y23[0] + y23[1])) * (1.0f / C);
                        d01 = y01 - pk1(mean);
                        d23 = y23 - pk1(mean);
                    } else if (pc == 2 && NORM2) {
                        var = ffw_row_sum<LPR>(
                            __builtin_fmaf(d23[1], d23[1], __builtin_fmaf(d23[0], d23[0], __builtin_fmaf(d01[1], d01[1], d01[0] * d01[0]))));
                    } else if (pc == 3 && NORM2) {
                        const float rstd = rsqrtf(var * (1.0f / C) + a.eps2);
                        const float4 ln2w = FFW_LN2W, ln2b = FFW_LN2B;
                        const f32x2 n01 = d01 * pk1(rstd) * pk2(ln2w.x, ln2w.y) + pk2(ln2b.x, ln2b.y),
                                    n23 = d23 * pk1(rstd) * pk2(ln2w.z, ln2w.w) + pk2(ln2b.z, ln2b.w);
                        *reinterpret_cast<float4 *>(a.y_norm + row * C + 4 * q) = make_float4(n01[0], n01[1], n23[0], n23[1]);
                    }
                }
                if (st == (NP + 1 < NS1 ? NP + 1 : NS1 - 1)) ffw_split4(FFW_U2(0, 0), FFW_U2(0, 2), h0_, l0_);
                __builtin_amdgcn_sched_barrier(0);
            }
            (void)NPC1;
            ffw_split4(FFW_U2(1, 0), FFW_U2(1, 2), h1_, l1_);
            ufrag[(wv * 2) * 64 + lane] = ffw_cat(h0_, h1_);
            ufrag[(wv * 2 + 1) * 64 + lane] = ffw_cat(l0_, l1_);
        }
        WSTAMP(4)
        __syncthreads();
        WSTAMP(5)
#ifdef MSSVT_STAMPS
        ++ws_tiles;
#endif
        xc = xn;
        if (!has_next) break;
        tile = tile_next;
    }
    WSTAMP(6)
#undef FFW_COMBINE
#undef FFW_NORM_TO_BFRAG
#undef FFW_GEMM1
#undef FFW_U
#undef FFW_U2
#ifdef MSSVT_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_ws_span[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && blockIdx.x < 4) {
        for (int k = 0; k < 7; ++k) g_ws_stamps[(blockIdx.x * 8 + (wv & 7)) * 16 + k] = ws_acc[k];
        g_ws_stamps[(blockIdx.x * 8 + (wv & 7)) * 16 + 9] = ws_tiles;
    }
#endif
#undef FFW_TAB
#undef FFW_ISSUE
}


template <int C, int FF>
static int launch_ffn_ws(const FfnArgs &a, const void *packed, hipStream_t stream) {
    constexpr int NW = FF / 32;
    const size_t lds = (size_t)(C / 32 + NW) * 2 * 64 * 16 + (size_t)16 * (C + 4) * 4 + (size_t)4 * C * 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int tiles = (a.n_rows + 15) / 16;
    int grid = cus * (NW >= 8 ? 1 : 8 / NW);  // 8 waves per CU
    if (grid > tiles) grid = tiles;
    if (grid < 1) return MSSVT_OK;
    const h16x8 *pk = reinterpret_cast<const h16x8 *>(packed);
    const dim3 block(NW * MSSVT_WAVE);
    if (a.tab_row && a.y_norm && !a.y) {
        k_ffn_ws<C, FF, true, true, false><<<grid, block, lds, stream>>>(a, pk);
    } else if (!a.y) {
        return MSSVT_E_BADARG;
    } else if (a.tab_row) {
        if (a.y_norm) k_ffn_ws<C, FF, true, true><<<grid, block, lds, stream>>>(a, pk);
        else k_ffn_ws<C, FF, true, false><<<grid, block, lds, stream>>>(a, pk);
    } else {
        if (a.y_norm) k_ffn_ws<C, FF, false, true><<<grid, block, lds, stream>>>(a, pk);
        else k_ffn_ws<C, FF, false, false><<<grid, block, lds, stream>>>(a, pk);
    }
    return mssvt_launch_status();
}

template <int C, int FF>
static int launch_ffn_pack(const float *W1, const float *W2, void *packed, hipStream_t stream) {
    k_ffn_pack<C, FF><<<FF / 32, MSSVT_WAVE, 0, stream>>>(W1, W2, reinterpret_cast<h16x8 *>(packed));
    return mssvt_launch_status();
}

extern "C" long long mssvt_ffn_packed_bytes(int C, int FF) {
    if (!((C == 128 && FF == 256) || (C == 64 && FF == 128) || (C == 32 && FF == 64))) return 0;
    return (long long)(FF / 32) * (4 * (C / 32) + 2 * (FF / 32)) * 64 * 16;
}

extern "C" int mssvt_ffn_pack_weights(int C, int FF, const float *W1, const float *W2, void *packed, void *stream) {
    if (!W1 || !W2 || !packed) return MSSVT_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (C == 128 && FF == 256) return launch_ffn_pack<128, 256>(W1, W2, packed, st);
    if (C == 64 && FF == 128) return launch_ffn_pack<64, 128>(W1, W2, packed, st);
    if (C == 32 && FF == 64) return launch_ffn_pack<32, 64>(W1, W2, packed, st);
    return MSSVT_E_TOOLARGE;
}

template <int C, int FF>
static int launch_ffn_split(const FfnArgs &a, float *hidden, int phases, hipStream_t stream) {
    const size_t lds_up = ((size_t)FF * (C + 4) + FF + 2 * C) * 4, lds_down = ((size_t)C * (FF + 4) + 3 * C) * 4;
    static_assert(((size_t)FF * (C + 4) + FF + 2 * C) * 4 <= 160 * 1024 && ((size_t)C * (FF + 4) + 3 * C) * 4 <= 160 * 1024,
                  "weights must fit the LDS");
    if (lds_up > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_ffn_up<C, FF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_up);
        if (e != hipSuccess) return (int)e;
    }
    if (lds_down > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_ffn_down<C, FF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_down);
        if (e != hipSuccess) return (int)e;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int tiles = (a.n_rows + 15) / 16;
    int grid = cus * (int)((160 * 1024) / (lds_up > lds_down ? lds_up : lds_down));
    if (grid > tiles) grid = tiles;
    if (grid < 1) return MSSVT_OK;
    if (phases & 1) k_ffn_up<C, FF><<<grid, FFS_NW * MSSVT_WAVE, lds_up, stream>>>(a, hidden);
    if (phases & 2) k_ffn_down<C, FF><<<grid, FFS_NW * MSSVT_WAVE, lds_down, stream>>>(a, hidden);
    return mssvt_launch_status();
}

static int dispatch_ffn(int C, int FF, const FfnArgs &a, float *hidden, int phases, hipStream_t st) {
    if (phases == 4) {  // single launch, split fp16 operands (the caller has checked their range); hidden = the
                        // fragments of mssvt_ffn_pack_weights, or NULL: split in the kernel's prologue
        if (C == 128 && FF == 256) return launch_ffn_ws<128, 256>(a, hidden, st);
        if (C == 64 && FF == 128) return launch_ffn_ws<64, 128>(a, hidden, st);
        if (C == 32 && FF == 64) return launch_ffn_ws<32, 64>(a, hidden, st);
        return MSSVT_E_TOOLARGE;
    }
    if (hidden) {
        if (C == 128 && FF == 256) return launch_ffn_split<128, 256>(a, hidden, phases, st);
        if (C == 64 && FF == 128) return launch_ffn_split<64, 128>(a, hidden, phases, st);
        if (C == 32 && FF == 64) return launch_ffn_split<32, 64>(a, hidden, phases, st);
        return MSSVT_E_TOOLARGE;
    }
    return MSSVT_E_BADARG;  // the fp32 form needs the (n_rows, FF) scratch
}

extern "C" int mssvt_ffn_fused(int n_rows, int C, int FF, const float *x_new, const float *x_in,
                               const int *owner, const float *norm_w, const float *norm_b, float eps,
                               const float *W1, const float *b1, const float *W2, const float *b2,
                               float *y, const float *next_norm_w, const float *next_norm_b,
                               float next_eps, float *y_norm, float *hidden, const int *num_rows_dev,
                               int phases, void *stream) {
    if (n_rows < 0 || !x_new || !norm_w || !norm_b || !W1 || !b1 || !W2 || !b2 || !y) return MSSVT_E_BADARG;
    if (owner && !x_in) return MSSVT_E_BADARG;
    if (y_norm && (!next_norm_w || !next_norm_b)) return MSSVT_E_BADARG;
    if (n_rows == 0) return MSSVT_OK;
    FfnArgs a;
    a.n_rows = n_rows; a.n_rows_dev = num_rows_dev; a.x_new = x_new; a.x_in = x_in; a.owner = owner;
    a.tab_row = nullptr; a.tab_w = nullptr; a.attn = nullptr;
    a.ln_w = norm_w; a.ln_b = norm_b; a.eps = eps;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.y = y;
    a.ln2_w = next_norm_w; a.ln2_b = next_norm_b; a.eps2 = next_eps; a.y_norm = y_norm; a.xcd = mssvt_xcd_remap(); a.prio = getenv("MSSVT_FFN_PRIO") ? atoi(getenv("MSSVT_FFN_PRIO")) : 1;
    return dispatch_ffn(C, FF, a, hidden, phases, (hipStream_t)stream);
}

// Same tail, fed by the interpolation table of mssvt_block_interp_table: the residual input
// x = x_in + sum_i w_i * attn[row_i] (or 2 * x_in for voxels no list slot owns) is built while the
// rows are loaded, so neither the scatter kernel nor its (N,C) output exist.
extern "C" int mssvt_ffn_fused_interp(int n_rows, int C, int FF, const float *x_in, const int *tab_row,
                                      const float *tab_w, const float *attn, const float *norm_w,
                                      const float *norm_b, float eps, const float *W1, const float *b1,
                                      const float *W2, const float *b2, float *y,
                                      const float *next_norm_w, const float *next_norm_b, float next_eps,
                                      float *y_norm, float *hidden, const int *num_rows_dev, int phases,
                                      void *stream) {
    if (n_rows < 0 || !x_in || !tab_row || !tab_w || !attn || !norm_w || !norm_b || !W1 || !b1 || !W2 || !b2)
        return MSSVT_E_BADARG;
    if (!y && !(phases == 4 && y_norm)) return MSSVT_E_BADARG;  // y may be NULL only for the single launch with a LayerNorm output
    if (y_norm && (!next_norm_w || !next_norm_b)) return MSSVT_E_BADARG;
    if (n_rows == 0) return MSSVT_OK;
    FfnArgs a;
    a.n_rows = n_rows; a.n_rows_dev = num_rows_dev; a.x_new = nullptr; a.x_in = x_in; a.owner = nullptr;
    a.tab_row = reinterpret_cast<const int4 *>(tab_row);
    a.tab_w = reinterpret_cast<const float4 *>(tab_w);
    a.attn = attn;
    a.ln_w = norm_w; a.ln_b = norm_b; a.eps = eps;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.y = y;
    a.ln2_w = next_norm_w; a.ln2_b = next_norm_b; a.eps2 = next_eps; a.y_norm = y_norm; a.xcd = mssvt_xcd_remap(); a.prio = getenv("MSSVT_FFN_PRIO") ? atoi(getenv("MSSVT_FFN_PRIO")) : 1;
    return dispatch_ffn(C, FF, a, hidden, phases, (hipStream_t)stream);
}
