// block_attn.hip -- fused mixed-scale window attention of one MsSVT Block (fp32).
//
// Replaces, for one head group g of one Block, the reference's
//   7 x K5 feature/coordinate gathers           (ref: mssvt_backbone.py:260-268)
//   relative coordinates + positional MLP        (ref: :269-282, pos_proj :43-47)
//   MixedScaleAttention.forward for group g      (ref: mssvt_utils.py:112-150)
// and, in the table kernel at the end of this file, K9 + K10 + the interpolation weights
// (ref: mssvt_backbone.py:298-311).  Nothing padded is written to HBM.
//
// Arithmetic is re-associated around the small side of the problem (#queries << #keys: ~2 valid
// queries against ~4 + ~20 unmasked keys per window at 160k points): with q' = Wq x_q + b_q,
//   score_h(k)  = scale q'_h . (Wk_h x_k + bk_h) = (scale Wk_h^T q'_h) . x_k + const_h
//   out_h       = sum_k p_hk (Wv_h x_k + bv_h)   = Wv_h (sum_k p_hk x_k) + bv_h
// (const_h cancels in the softmax, sum_k p_hk = 1): keys are never projected; per query 4 mat-vecs
// of size Cg^2, per (query,key) pair 2*heads*Cg MACs.  Masked key slots (additive -100 in the
// reference -> relative weight <= e^-100) are skipped; slot 0 of each scale is never masked, so
// no key set is empty.  Differences to the reference are re-association only (~1e-6 relative).
//
// THREE LAUNCHES:
//   A  k_attn_q  : rows = the compact list of valid queries (mssvt_plan_order), 16 per wavefront:
//                  x_q = xhat row + pos. embedding; Q' = X_q Wq^T + b; Qt_h = scale Q'_h Wk_h -> qbuf
//   B  k_attn_kv : one wavefront per window, lane = channel: key tokens (xhat rows + pos. embedding)
//                  -> LDS; per query scores, softmax, xbar_h = sum_k p_hk x_k -> qbuf (in place of Qt)
//   C  k_attn_o  : rows again: V = Xbar_{head} Wv^T + bv; out = V Wo^T + bo -> attn rows
// A single fused kernel (first version) needs all four Cg x Cg matrices (64 KiB) PLUS the key tile
// (9 KiB per wave) in LDS: 7-8 waves per CU, every LDS / DPP / gather latency exposed (all pipes
// ~25 % busy, 190 us per launch); and a per-query mat-vec re-reads a whole matrix from LDS per query.
// Split, B holds no weights at all (14 waves per CU), and A / C are plain row-tiled GEMMs on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products, fp32 accumulation): weights are read
// from LDS once per 16 queries.  The price is one 4*HP*Cg-byte row per query and group written by A,
// rewritten by B, read by C (qbuf, L2/MALL resident).
//
// MFMA mapping of A and C (lane l: r = l % 16 = query row of the tile, g = l / 16): every GEMM is
// computed TRANSPOSED, D^T[out][row] = sum_k W[out][k] X[row][k], A operand = weight rows (one
// ds_read_b128 of the natural row-major matrix feeds 4 k-steps: k-step (S, j) <-> column 16 S + 4 g + j),
// B operand = the activations.  The accumulator layout (lane (r, g), register i = output 4 g + i of
// the 16-tile) then IS the B operand of the next GEMM, step by step, with no shuffle or LDS round
// trip; and it is 4 consecutive channels of one row = one 16-byte global load / store.
// LDS rows are padded to Cg + 4 floats: the 16 lanes of a b128 phase hit 16 x 4 distinct banks.
//
// Windows (B) are processed in the plan's work order (heaviest first), dealt round-robin to the
// wavefronts of a persistent grid.
#include <stdlib.h>
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ATTN_ROW_WAVES 4  // waves per workgroup of the per-window launch
// waves per workgroup of the row-tiled launches (q, o): ONE workgroup per CU and head group -- the weights are staged
// once per CU (two Cg x Cg matrices, 37 KiB) instead of once per 4-wave workgroup (four of them per CU: 148 KiB at the
// ~11 B/clk a CU fills at = the larger part of a launch on 8k query rows); 66-71 VGPRs leave room for 4 waves / SIMD
#define ATTN_QO_WAVES 16
#define AST_UN 16        // weight elements per matrix and thread in flight while staging (64 x 64 / 256 threads)

struct AttnArgs {
    int C, c0, heads, hd;
    float scale;
    int nq, K;
    const float *xhat;
    const int *num_wins;  // number of entries of `perm` (windows with at least one valid query)
    const int *perm;      // work order: heavy windows first
    const int *q_off;     // (cap) first compact query row of each window
    const int *nq_valid;  // (cap) valid queries of each window
    // compact query rows (mssvt_plan_order): (rel.x, rel.y, rel.z, bits(global feature row));
    // (window, attn row)
    const int *num_rows;
    const float4 *qrow_meta;
    const int2 *qrow_src;
    // per-key-slot metadata resolved by the plan kernel (window_plan.hip): (rel.x, rel.y, rel.z,
    // bits(global feature row or -1)); wcentre = window centre in metres
    const float4 *kmeta, *wcentre;
    const float *Wq, *bq, *Wkv, *bkv, *Wo, *bo, *Wp, *bp;
    float *qbuf;  // (query rows, HP*CG): Qt after A, Xbar after B
    float *attn;
    int row_capacity;  // rows of qbuf / the compact row arrays
    const void *packed;  // split-fp16 weight fragments of this group (mssvt_attn_pack_weights) or null
    int xcd;  // deal the work order so that an XCD owns a contiguous run per round (common.hip.h, xcd_contiguous_block)
};

// head groups of equal shape run in ONE launch: blockIdx.y = group (their work is independent: channel
// slices of the same rows); a light group fills the gaps of a heavy one and launches / tails halve
#define ATTN_MAX_GROUPS 4
struct AttnPack {
    AttnArgs g[ATTN_MAX_GROUPS];
};

#define MFMA4(acc, av, bv)                                                    \
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av), (bv), acc, 0, 0, 0)

// ---- split-fp16 operands (kv16 form of launch B, see k_attn_kvh) ---------------------------------------------
// v = hi + 2^-11 lo with hi = fp16(v), lo = fp16((v - hi) 2^11), both rounded toward zero: 22 mantissa bits; a product
// sum is three v_mfma_f32_16x16x32_f16 (hi hi, hi lo, lo hi) accumulated in fp32 -- the arithmetic of csrc/ffn.hip
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16v4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define MFMA_H(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16((av), (bv), acc, 0, 0, 0)
#define H16_SCALE 2048.0f
#define H16_INV (1.0f / 2048.0f)
__device__ __forceinline__ void h16_split4(const f32x4 v, h16x4 &hi, h16x4 &lo) {
    const fp16x2 a = __builtin_amdgcn_cvt_pkrtz(v[0], v[1]), b = __builtin_amdgcn_cvt_pkrtz(v[2], v[3]);
    // (v - hi) 2^11 as fma(hi, -2^11, v 2^11): every step is exact, so the bits are those of the subtraction form, and
    // the fp16 -> fp32 conversion of hi rides inside the instruction (v_fma_mix_f32): 3 instead of 4 VALU per value
    const f32x2 s01 = f32x2{v[0], v[1]} * f32x2{H16_SCALE, H16_SCALE}, s23 = f32x2{v[2], v[3]} * f32x2{H16_SCALE, H16_SCALE};  // v_pk_mul_f32
    const fp16x2 c = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)a[0], -H16_SCALE, s01[0]), __builtin_fmaf((float)a[1], -H16_SCALE, s01[1])),
                 d = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)b[0], -H16_SCALE, s23[0]), __builtin_fmaf((float)b[1], -H16_SCALE, s23[1]));
    hi = h16x4{(_Float16)a[0], (_Float16)a[1], (_Float16)b[0], (_Float16)b[1]};
    lo = h16x4{(_Float16)c[0], (_Float16)c[1], (_Float16)d[0], (_Float16)d[1]};
}
__device__ __forceinline__ h16x8 h16_cat(const h16x4 a, const h16x4 b) {
    return h16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// (hi, lo) fragments of 8 values held as two accumulator-layout quads
__device__ __forceinline__ void h16_split8(const f32x4 v0, const f32x4 v1, h16x8 &hi, h16x8 &lo) {
    h16x4 h0, l0, h1, l1;
    h16_split4(v0, h0, l0);
    h16_split4(v1, h1, l1);
    hi = h16_cat(h0, h1);
    lo = h16_cat(l0, l1);
}
// kv16 operand order: k slot (g, j) of a 32-wide step P <-> index 32 P + 16 (j / 4) + 4 g + j % 4 -- what a lane holds after
// two DENSE 16-byte row pieces S = 2 P, 2 P + 1 (piece S of lane (., g) = indices 16 S + 4 g .. + 3: the four g lanes of a
// row read 64 contiguous bytes), and equally the accumulator registers of two consecutive 16-row output tiles

// Store of the Q~ / Xbar hand-off rows.  (Measured: written through with `sc1` instead of left dirty for the
// write-back at the kernel boundary, the 67 MB per launch cost the same ~11 us -- k_attn_q 20.6 -> 19.2 us,
// k_attn_kv 46.0 -> 47.2 us: it is the bytes, not when they leave the L2.)
__device__ __forceinline__ void store_handoff(float *p, f32x4 v) { *reinterpret_cast<f32x4 *>(p) = v; }

#ifdef MSSVT_STAMPS
__device__ unsigned long long g_attn_q_stamps[64 * 8];
extern "C" int mssvt_debug_read_attn_q_stamps(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_q_stamps), sizeof(g_attn_q_stamps));
}
#define QSTAMP(k_) if (threadIdx.x == 0 && blockIdx.x < 64 && blockIdx.y == 0) g_attn_q_stamps[blockIdx.x * 8 + (k_)] = __builtin_readcyclecounter();
#else
#define QSTAMP(k_)
#endif
// ---- A: queries -> Qt --------------------------------------------------------------------------
// KV16: Qt rows leave as (hi, lo) fp16 fragments in the operand order of k_attn_kvh (same bytes per row): per head,
// [step P][g][hi x 8 | lo x 8], the 8 = this lane's accumulator registers of output tiles 2 P and 2 P + 1
template <int CG, int HD, int HP, bool KV16>
__global__ void __launch_bounds__(ATTN_QO_WAVES *MSSVT_WAVE) k_attn_q(AttnPack pack) {
    static_assert(!KV16 || CG % 32 == 0, "kv16 operand order: 32-channel steps");
    const AttnArgs &a = pack.g[blockIdx.y];
    constexpr int CGP = (CG + 15) / 16 * 16, NT = CGP / 16, LS = CGP + 4, NH = CG / HD, QROW = HP * CG;
    extern __shared__ float4 lds4[];
    float *Wq_l = reinterpret_cast<float *>(lds4);  // [o][c]   natural nn.Linear layout, padded rows
    float *WkT_l = Wq_l + CGP * LS;                 // [c][o] = Wk[o][c]
    float *Wp_l = WkT_l + CGP * LS;                 // [c][8] = pos_proj row (6 weights, bias, 0)
    float *bq_l = Wp_l + CGP * 8;                   // [o]
    // tile k * grid + block -> wave k % waves: a short row list is spread over all CUs; a workgroup without a tile
    // leaves before it stages anything
    QSTAMP(0)
    const int lane = lane_id(), r = lane & 15, g = lane >> 4;
    const int rows = *a.num_rows, tiles = (rows + 15) >> 4;
    const int wv = threadIdx.x / MSSVT_WAVE;
    if ((int)blockIdx.x >= tiles) return;
    QSTAMP(1)
    // the first tile's row metadata travels while the weights are staged, its feature rows while the workgroup
    // meets at the barrier: metadata -> rows -> products is a chain of round trips as long as the staging itself
    int tile = wv * gridDim.x + blockIdx.x;
    float4 rm = a.qrow_meta[min(tile * 16 + r, rows - 1)];
    int2 src = a.qrow_src[min(tile * 16 + r, rows - 1)];
    // AST_UN elements of each matrix in flight per thread (a plain copy loop waits for every load before
    // its store: one global round trip per element)
    for (int e0 = threadIdx.x; e0 < CGP * CGP; e0 += blockDim.x * AST_UN) {
        float vq[AST_UN], vk[AST_UN];
#pragma unroll
        for (int u = 0; u < AST_UN; ++u) {
            const int e = e0 + u * blockDim.x, o = e / CGP, c = e % CGP;
            const bool in = e < CGP * CGP && o < CG && c < CG;
            vq[u] = in ? a.Wq[o * CG + c] : 0.f;
            vk[u] = in ? a.Wkv[o * CG + c] : 0.f;  // rows [0,CG) of to_kvs = K projection
        }
#pragma unroll
        for (int u = 0; u < AST_UN; ++u) {
            const int e = e0 + u * blockDim.x, o = e / CGP, c = e % CGP;
            if (e < CGP * CGP) {
                Wq_l[o * LS + c] = vq[u];
                WkT_l[c * LS + o] = vk[u];
            }
        }
    }
    for (int e = threadIdx.x; e < CGP * 8; e += blockDim.x) {
        const int c = e >> 3, t = e & 7;
        Wp_l[e] = c < CG ? (t < 6 ? a.Wp[(size_t)(a.c0 + c) * 6 + t] : (t == 6 ? a.bp[a.c0 + c] : 0.f)) : 0.f;
    }
    for (int e = threadIdx.x; e < CGP; e += blockDim.x) bq_l[e] = e < CG ? a.bq[e] : 0.f;
    QSTAMP(2)
    float4 wc = a.wcentre[src.x];
    f32x4 xq[NT];
#define ATTN_Q_ROWS()                                                                                     \
    {                                                                                                     \
        const float *xrow_ = a.xhat + (size_t)__builtin_bit_cast(int, rm.w) * a.C + a.c0;                 \
        _Pragma("unroll") for (int S = 0; S < NT; ++S) {                                                  \
            const int c_ = 16 * S + 4 * g;                                                                \
            const float4 v_ = (CGP == CG || c_ < CG) ? *reinterpret_cast<const float4 *>(xrow_ + c_)      \
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);                   \
            xq[S] = f32x4{v_.x, v_.y, v_.z, v_.w};                                                        \
        }                                                                                                 \
    }
    ATTN_Q_ROWS()
    __syncthreads();
    QSTAMP(3)
    for (bool first = true; tile < tiles; tile += gridDim.x * ATTN_QO_WAVES, first = false) {
        const int row = min(tile * 16 + r, rows - 1);
        if (!first) {
            rm = a.qrow_meta[row];
            src = a.qrow_src[row];
            wc = a.wcentre[src.x];
            // B operand of GEMM1: this lane's 4 channels 16 S + 4 g + j of its query token
            ATTN_Q_ROWS()
        }
#pragma unroll
        for (int S = 0; S < NT; ++S) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 p0 = *reinterpret_cast<const float4 *>(Wp_l + (16 * S + 4 * g + j) * 8);
                const float4 p1 = *reinterpret_cast<const float4 *>(Wp_l + (16 * S + 4 * g + j) * 8 + 4);
                const float posc = p1.z + p0.w * wc.x + p1.x * wc.y + p1.y * wc.z;  // window part of the pos. MLP
                xq[S][j] += fmaxf(posc + p0.x * rm.x + p0.y * rm.y + p0.z * rm.z, 0.0f);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads of later steps from being hoisted (VGPRs)
        }
        // GEMM1^T: Q'^T[o][row] = sum_c Wq[o][c] xq[row][c] + bq[o]
        f32x4 qp[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 b = *reinterpret_cast<const float4 *>(bq_l + 16 * t + 4 * g);
            qp[t] = f32x4{b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int S = 0; S < NT; ++S) {
            float4 w[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) w[t] = *reinterpret_cast<const float4 *>(Wq_l + (16 * t + r) * LS + 16 * S + 4 * g);
#pragma unroll
            for (int t = 0; t < NT; ++t) MFMA4(qp[t], w[t].x, xq[S][0]);
#pragma unroll
            for (int t = 0; t < NT; ++t) MFMA4(qp[t], w[t].y, xq[S][1]);
#pragma unroll
            for (int t = 0; t < NT; ++t) MFMA4(qp[t], w[t].z, xq[S][2]);
#pragma unroll
            for (int t = 0; t < NT; ++t) MFMA4(qp[t], w[t].w, xq[S][3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // GEMM2^T per head: Qt_h^T[c][row] = scale sum_{o in head h} Wk[o][c] Q'[row][o]
        const bool row_ok = tile * 16 + r < rows;
        float *dst = a.qbuf + (size_t)(tile * 16 + r) * QROW;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            f32x4 acc[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (16 * t >= (h + 1) * HD || 16 * (t + 1) <= h * HD) continue;  // tile outside this head
                // 4 consecutive outputs 16 t + 4 g + i of Q' belong to one head (HD % 4 == 0)
                const bool mine = (HD % 16 == 0) || (16 * t + 4 * g) / HD == h;
                const f32x4 bq4 = mine ? qp[t] : f32x4{0.f, 0.f, 0.f, 0.f};
                float4 w[NT];
#pragma unroll
                for (int u = 0; u < NT; ++u) w[u] = *reinterpret_cast<const float4 *>(WkT_l + (16 * u + r) * LS + 16 * t + 4 * g);
#pragma unroll
                for (int u = 0; u < NT; ++u) MFMA4(acc[u], w[u].x, bq4[0]);
#pragma unroll
                for (int u = 0; u < NT; ++u) MFMA4(acc[u], w[u].y, bq4[1]);
#pragma unroll
                for (int u = 0; u < NT; ++u) MFMA4(acc[u], w[u].z, bq4[2]);
#pragma unroll
                for (int u = 0; u < NT; ++u) MFMA4(acc[u], w[u].w, bq4[3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (KV16) {
                h16x8 *dsth = reinterpret_cast<h16x8 *>(dst + h * CG);
#pragma unroll
                for (int P = 0; P < NT / 2; ++P) {
                    h16x8 hi, lo;
                    h16_split8(acc[2 * P] * a.scale, acc[2 * P + 1] * a.scale, hi, lo);
                    if (row_ok) {
                        dsth[(P * 4 + g) * 2] = hi;
                        dsth[(P * 4 + g) * 2 + 1] = lo;
                    }
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int c = 16 * u + 4 * g;
                if (row_ok && (CGP == CG || c < CG))
                    store_handoff(dst + h * CG + c,
                                  f32x4{acc[u][0] * a.scale, acc[u][1] * a.scale, acc[u][2] * a.scale, acc[u][3] * a.scale});
            }
        }
    }
    QSTAMP(4)
}

// ---- C: Xbar -> attention output rows ----------------------------------------------------------
template <int CG, int HD, int HP>
__global__ void __launch_bounds__(ATTN_QO_WAVES *MSSVT_WAVE) k_attn_o(AttnPack pack) {
    const AttnArgs &a = pack.g[blockIdx.y];
    constexpr int CGP = (CG + 15) / 16 * 16, NT = CGP / 16, LS = CGP + 4, NH = CG / HD, QROW = HP * CG;
    extern __shared__ float4 lds4[];
    float *Wv_l = reinterpret_cast<float *>(lds4);  // [o][c]
    float *Wo_l = Wv_l + CGP * LS;                  // [p][o]
    float *bv_l = Wo_l + CGP * LS, *bo_l = bv_l + CGP;
    const int lane = lane_id(), r = lane & 15, g = lane >> 4;
    const int rows = *a.num_rows, tiles = (rows + 15) >> 4;
    const int wv = threadIdx.x / MSSVT_WAVE;
    if ((int)blockIdx.x >= tiles) return;
    // the first tile's Xbar rows of head 0 travel while the weights are staged
    int tile = wv * gridDim.x + blockIdx.x;
    f32x4 x[NT], xn[NT];
#define ATTN_O_ROWS(dst_, row_, h_)                                                                       \
    {                                                                                                     \
        const float *xb_ = a.qbuf + (size_t)(row_) * QROW + (h_) * CG;                                    \
        _Pragma("unroll") for (int S = 0; S < NT; ++S) {                                                  \
            const int c_ = 16 * S + 4 * g;                                                                \
            const float4 t4_ = (CGP == CG || c_ < CG) ? *reinterpret_cast<const float4 *>(xb_ + c_)       \
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);                  \
            dst_[S] = f32x4{t4_.x, t4_.y, t4_.z, t4_.w};                                                  \
        }                                                                                                 \
    }
    ATTN_O_ROWS(x, min(tile * 16 + r, rows - 1), 0)
    for (int e0 = threadIdx.x; e0 < CGP * CGP; e0 += blockDim.x * AST_UN) {
        float vv[AST_UN], vo[AST_UN];
#pragma unroll
        for (int u = 0; u < AST_UN; ++u) {
            const int e = e0 + u * blockDim.x, o = e / CGP, c = e % CGP;
            const bool in = e < CGP * CGP && o < CG && c < CG;
            vv[u] = in ? a.Wkv[(size_t)(CG + o) * CG + c] : 0.f;  // rows [CG,2CG) of to_kvs = V projection
            vo[u] = in ? a.Wo[o * CG + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < AST_UN; ++u) {
            const int e = e0 + u * blockDim.x, o = e / CGP, c = e % CGP;
            if (e < CGP * CGP) {
                Wv_l[o * LS + c] = vv[u];
                Wo_l[o * LS + c] = vo[u];
            }
        }
    }
    for (int e = threadIdx.x; e < CGP; e += blockDim.x) {
        bv_l[e] = e < CG ? a.bkv[CG + e] : 0.f;
        bo_l[e] = e < CG ? a.bo[e] : 0.f;
    }
    __syncthreads();
    for (bool first = true; tile < tiles; tile += gridDim.x * ATTN_QO_WAVES, first = false) {
        const int row = min(tile * 16 + r, rows - 1);
        const bool row_ok = tile * 16 + r < rows;
        const int dest = a.qrow_src[row].y;
        if (!first) ATTN_O_ROWS(x, row, 0)
        // GEMM3^T: V^T[o][row] = sum_c Wv[o][c] Xbar_{head(o)}[row][c] + bv[o]
        f32x4 v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 b = *reinterpret_cast<const float4 *>(bv_l + 16 * t + 4 * g);
            v[t] = f32x4{b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            if (h + 1 < NH) ATTN_O_ROWS(xn, row, h + 1)  // the next head's rows under this head's products
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (16 * t >= (h + 1) * HD || 16 * (t + 1) <= h * HD) continue;
                // A rows o = 16 t + r outside head h contribute nothing to this head's pass
                const bool mine = (HD % 16 == 0) || (16 * t + r) / HD == h;
#pragma unroll
                for (int S = 0; S < NT; ++S) {
                    float4 w = *reinterpret_cast<const float4 *>(Wv_l + (16 * t + r) * LS + 16 * S + 4 * g);
                    if (!mine) w = make_float4(0.f, 0.f, 0.f, 0.f);
                    MFMA4(v[t], w.x, x[S][0]);
                    MFMA4(v[t], w.y, x[S][1]);
                    MFMA4(v[t], w.z, x[S][2]);
                    MFMA4(v[t], w.w, x[S][3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int S = 0; S < NT; ++S) x[S] = xn[S];
        }
        // GEMM4^T: out^T[p][row] = sum_o Wo[p][o] V[row][o] + bo[p]
        f32x4 out[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float4 b = *reinterpret_cast<const float4 *>(bo_l + 16 * u + 4 * g);
            out[u] = f32x4{b.x, b.y, b.z, b.w};
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float4 w[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) w[u] = *reinterpret_cast<const float4 *>(Wo_l + (16 * u + r) * LS + 16 * t + 4 * g);
#pragma unroll
            for (int u = 0; u < NT; ++u) MFMA4(out[u], w[u].x, v[t][0]);
#pragma unroll
            for (int u = 0; u < NT; ++u) MFMA4(out[u], w[u].y, v[t][1]);
#pragma unroll
            for (int u = 0; u < NT; ++u) MFMA4(out[u], w[u].z, v[t][2]);
#pragma unroll
            for (int u = 0; u < NT; ++u) MFMA4(out[u], w[u].w, v[t][3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        float *dst = a.attn + (size_t)dest * a.C + a.c0;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int c = 16 * u + 4 * g;
            if (row_ok && (CGP == CG || c < CG))
                *reinterpret_cast<float4 *>(dst + c) = make_float4(out[u][0], out[u][1], out[u][2], out[u][3]);
        }
    }
}

// ---- A and C with split-fp16 operands (kv16 form) ---------------------------------------------------------------
// The row-tiled launches above are bound by the fp32 matrix instruction too: 128 v_mfma_f32_16x16x4_f32 = 4 096 pipe cycles
// per 16-row tile, four waves per SIMD.  Here the four Cg x Cg matrices are split ONCE per parameter version into (hi, lo)
// fp16 fragments in MFMA operand order (k_attn_pack; the softmax scale folded into Wk), staged by a straight 16-byte copy
// (32 KiB per launch and workgroup instead of 37 KiB of element-wise transposes), and a tile costs 72 (A) / 48 (C)
// sixteen-cycle instructions.  Workgroups are 8 waves so that BOTH head groups of a launch are resident side by side
// (the 16-wave form ran its two groups one after the other: 90 VGPRs allow one such workgroup per CU).
// Fragment blob of one head group (bytes; NT = Cg / 16 output tiles, NP = Cg / 32 steps; HD = 16: head h = tile h):
//   WqF [t][P][hi | lo][lane] x 16 B   A rows o = 16 t + m,        k slot (g, j) <-> channel 32 P + 16 (j / 4) + 4 g + j % 4
//   WkF [h][u][hi | lo][lane] x  8 B   A rows c = (16 * u + m), k slot (g, j) <-> o = 16 h + 4 g + j   (x scale)
//   WvF [t][P][hi | lo][lane] x 16 B   A rows o = 16 t + m,        k slot (g, j) <-> the same channel of Xbar_t
//   WoF [u][s][hi | lo][lane] x 16 B   A rows p = 16 u + m,        k slot (g, j) <-> o = 32 s + 16 (j / 4) + 4 g + j % 4
//   WkF2 [p][u][hi | lo][lane] x 16 B  WkF (x log2 e) of heads 2 p (j < 4) and 2 p + 1 (j >= 4) side by side: one K = 32 instruction
//                                      per head PAIR in k_attn_kvh<.., QP> (its B operand is zero outside the column's head)
#define ATTN_QO16_WAVES 8
#define MFMA_H16(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x16f16((av), (bv), acc, 0, 0, 0)
template <int CG>
struct AttnBlob {
    static constexpr int NT = CG / 16, NP = CG / 32;
    static constexpr int WQ = 0, WK = WQ + NT * NP * 2 * 64 * 16, WV = WK + NT * NT * 2 * 64 * 8, WO = WV + NT * NP * 2 * 64 * 16,
                         WK2 = WO + NT * NP * 2 * 64 * 16, WV2 = WK2 + (NT / 2) * NT * 2 * 64 * 16,
                         BYTES = WV2 + NT * NP * 2 * 64 * 16;
    // WvF2 [t][P][hi | lo][lane] x 16 B (round 6): Wv for the window launch -- A rows o = 16 t + m, k slot (g, j) <-> channel
    // 32 P + 16 (g % 2) + 8 (j / 4) + 4 (g / 2) + j % 4 = the order in which k_attn_kvh's Xbar accumulators sit in its lanes
};

template <int CG>
__global__ void __launch_bounds__(MSSVT_WAVE) k_attn_pack(const float *Wq, const float *Wkv, const float *Wo, float scale, char *blob) {
    using L = AttnBlob<CG>;
    constexpr int NT = L::NT, NP = L::NP;
    const int lane = lane_id(), m = lane & 15, g = lane >> 4, f = blockIdx.x;  // fragment index within its matrix
    const int which = blockIdx.y;                                              // 0 Wq, 1 Wk, 2 Wv, 3 Wo, 4 Wk head pairs, 5 Wv (window launch)
    if (which == 1) {
        if (f >= NT * NT) return;
        const int h = f / NT, u = f % NT;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = Wkv[(size_t)(16 * h + 4 * g + j) * CG + (16 * u + m)] * scale;
        h16x4 hi, lo;
        h16_split4(v, hi, lo);
        h16x4 *dst = reinterpret_cast<h16x4 *>(blob + L::WK) + (size_t)f * 2 * 64 + lane;
        dst[0] = hi;
        dst[64] = lo;
        return;
    }
    if (which == 4) {
        if (f >= (NT / 2) * NT) return;
        const int pr = f / NT, u = f % NT;
        f32x4 v0, v1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // (x log2 e: k_attn_kvh<.., QP> takes its softmax in base 2)
            v0[j] = Wkv[(size_t)(16 * (2 * pr) + 4 * g + j) * CG + (16 * u + m)] * (scale * 1.4426950408889634f);
            v1[j] = Wkv[(size_t)(16 * (2 * pr + 1) + 4 * g + j) * CG + (16 * u + m)] * (scale * 1.4426950408889634f);
        }
        h16x8 hi, lo;
        h16_split8(v0, v1, hi, lo);
        h16x8 *dst = reinterpret_cast<h16x8 *>(blob + L::WK2) + (size_t)f * 2 * 64 + lane;
        dst[0] = hi;
        dst[64] = lo;
        return;
    }
    if (f >= NT * NP) return;
    const int t = f / NP, P = f % NP;
    if (which == 5) {  // Wv in the lane order of k_attn_kvh's Xbar tiles (AttnBlob::WV2)
        const float *rowp = Wkv + (size_t)(CG + 16 * t + m) * CG + 32 * P + 16 * (g & 1) + 4 * (g >> 1);
        const float4 w0 = *reinterpret_cast<const float4 *>(rowp), w1 = *reinterpret_cast<const float4 *>(rowp + 8);
        h16x8 hi, lo;
        h16_split8(f32x4{w0.x, w0.y, w0.z, w0.w}, f32x4{w1.x, w1.y, w1.z, w1.w}, hi, lo);
        h16x8 *dst = reinterpret_cast<h16x8 *>(blob + L::WV2) + (size_t)f * 2 * 64 + lane;
        dst[0] = hi;
        dst[64] = lo;
        return;
    }
    // 8 k slots of a lane = two runs of 4 consecutive columns, 16 apart (kv16 operand order)
    const float *rowp = (which == 0 ? Wq + (size_t)(16 * t + m) * CG : which == 2 ? Wkv + (size_t)(CG + 16 * t + m) * CG
                                                                                  : Wo + (size_t)(16 * t + m) * CG) + 32 * P + 4 * g;
    const float4 w0 = *reinterpret_cast<const float4 *>(rowp), w1 = *reinterpret_cast<const float4 *>(rowp + 16);
    h16x8 hi, lo;
    h16_split8(f32x4{w0.x, w0.y, w0.z, w0.w}, f32x4{w1.x, w1.y, w1.z, w1.w}, hi, lo);
    h16x8 *dst = reinterpret_cast<h16x8 *>(blob + (which == 0 ? L::WQ : which == 2 ? L::WV : L::WO)) + (size_t)f * 2 * 64 + lane;
    dst[0] = hi;
    dst[64] = lo;
}

// straight copy of n16 16-byte pieces into the LDS, every load of a thread in flight before its first store
template <int N16, int THREADS>
__device__ __forceinline__ void attn_stage16(float4 *lds, const float4 *src) {
    constexpr int PER = (N16 + THREADS - 1) / THREADS;
    float4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = threadIdx.x + k * THREADS;
        if (N16 % THREADS == 0 || e < N16) v[k] = src[e];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = threadIdx.x + k * THREADS;
        if (N16 % THREADS == 0 || e < N16) lds[e] = v[k];
    }
}

// ---- A, kv16 form: rows -> Qt.  OUT 0: Qt in fp32 for k_attn_kv; 1: Qt as (hi, lo) fragments for k_attn_kvh<.., false>;
// 2: only Q' (scaled by the Wk fragments later) as (hi, lo) B fragments of the K = 16 product, [head][g][hi x 4 | lo x 4]
// = 4 Cg bytes at the start of the row -- k_attn_kvh<.., true> multiplies by Wk_h itself: a quarter of the hand-off bytes
// (this launch is bound by writing them: 67 MB for 33 k query rows) and of this launch's matrix work
// (sched_barriers: without them the scheduler hoists the fragment reads of every step and spills ~100 registers)
template <int CG, int HP, int OUT>
__global__ void __launch_bounds__(ATTN_QO16_WAVES *MSSVT_WAVE, 4) k_attn_q16(AttnPack pack) {
    const AttnArgs &a = pack.g[blockIdx.y];
    using L = AttnBlob<CG>;
    constexpr int NT = CG / 16, NP = CG / 32, QROW = HP * CG;
    extern __shared__ float4 lds4[];
    const h16x8 *WqF = reinterpret_cast<const h16x8 *>(lds4);
    const h16x4 *WkF = reinterpret_cast<const h16x4 *>(reinterpret_cast<const char *>(lds4) + L::WK);
    constexpr int STAGED = OUT == 2 ? L::WK : L::WV;  // Wq fragments (+ Wk fragments)
    float *bq_l = reinterpret_cast<float *>(reinterpret_cast<char *>(lds4) + STAGED);
    const int lane = lane_id(), r = lane & 15, g = lane >> 4;
    const int rows = *a.num_rows, tiles = (rows + 15) >> 4;
    const int wv = threadIdx.x / MSSVT_WAVE;
    if ((int)blockIdx.x >= tiles) return;
    int tile = wv * gridDim.x + blockIdx.x;
    float4 rm = a.qrow_meta[min(tile * 16 + r, rows - 1)];
    int2 src = a.qrow_src[min(tile * 16 + r, rows - 1)];
    attn_stage16<STAGED / 16, ATTN_QO16_WAVES * MSSVT_WAVE>(lds4, reinterpret_cast<const float4 *>(a.packed));
    if (threadIdx.x < CG) bq_l[threadIdx.x] = a.bq[threadIdx.x];
    // positional MLP as two K = 4 products (relative offset | 1, window centre): A row r of tile u <-> channel (16 * u + r)
    float wrel[NT], wctr[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const float *wp = a.Wp + (size_t)(a.c0 + (16 * u + r)) * 6;
        wrel[u] = g < 3 ? wp[g] : a.bp[a.c0 + (16 * u + r)];
        wctr[u] = g < 3 ? wp[3 + g] : 0.f;
    }
    float4 wc = a.wcentre[src.x];
    f32x4 xq[NT];  // piece S: channels 16 S + 4 g + i of the row
#define ATTN_Q16_ROWS()                                                                                   \
    {                                                                                                     \
        const float *xrow_ = a.xhat + (size_t)__builtin_bit_cast(int, rm.w) * a.C + a.c0 + 4 * g;         \
        _Pragma("unroll") for (int S = 0; S < NT; ++S) {                                                  \
            const float4 v_ = *reinterpret_cast<const float4 *>(xrow_ + 16 * S);                          \
            xq[S] = f32x4{v_.x, v_.y, v_.z, v_.w};                                                        \
        }                                                                                                 \
    }
    ATTN_Q16_ROWS()
    __syncthreads();
    for (bool first = true; tile < tiles; tile += gridDim.x * ATTN_QO16_WAVES, first = false) {
        const int row = min(tile * 16 + r, rows - 1);
        if (!first) {
            rm = a.qrow_meta[row];
            src = a.qrow_src[row];
            wc = a.wcentre[src.x];
            ATTN_Q16_ROWS()
        }
        const float relb = lane_pick4(g, rm.x, rm.y, rm.z, 1.0f);
        const float ctrb = lane_pick4(g, wc.x, wc.y, wc.z, 0.0f);
        h16x8 xh[NP], xl[NP];
#pragma unroll
        for (int P = 0; P < NP; ++P) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 p1 = f32x4{0.f, 0.f, 0.f, 0.f};
                MFMA4(p1, wrel[2 * P + h], relb);
                MFMA4(p1, wctr[2 * P + h], ctrb);
#pragma unroll
                for (int i = 0; i < 4; ++i) xq[2 * P + h][i] += fmaxf(p1[i], 0.0f);
            }
            h16_split8(xq[2 * P], xq[2 * P + 1], xh[P], xl[P]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // GEMM1^T: Q'^T[o][row] = sum_c Wq[o][c] xq[row][c] + bq[o]; accumulator (row, g), i <-> o = 16 t + 4 g + i
        h16x4 qh[NT], ql[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 b = *reinterpret_cast<const float4 *>(bq_l + 16 * t + 4 * g);
            f32x4 mm = f32x4{b.x, b.y, b.z, b.w}, cr = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int P = 0; P < NP; ++P) {
                const h16x8 wh = WqF[((t * NP + P) * 2) * 64 + lane], wl = WqF[((t * NP + P) * 2 + 1) * 64 + lane];
                MFMA_H(mm, wh, xh[P]);
                MFMA_H(cr, wh, xl[P]);
                MFMA_H(cr, wl, xh[P]);
            }
            h16_split4(f32x4{__builtin_fmaf(cr[0], H16_INV, mm[0]), __builtin_fmaf(cr[1], H16_INV, mm[1]),
                             __builtin_fmaf(cr[2], H16_INV, mm[2]), __builtin_fmaf(cr[3], H16_INV, mm[3])}, qh[t], ql[t]);
            if (t & 1) __builtin_amdgcn_sched_barrier(0);
        }
        const bool row_ok = tile * 16 + r < rows;
        float *dst = a.qbuf + (size_t)(tile * 16 + r) * QROW;
        if (OUT == 2) {  // lane (row, g) holds Q'[row][16 t + 4 g + i]: the B fragment of head t, lanes (., g)
            h16x8 *dq = reinterpret_cast<h16x8 *>(dst);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (row_ok) dq[t * 4 + g] = h16_cat(qh[t], ql[t]);
            continue;
        }
        // GEMM2^T per head (= tile h): Qt_h^T[c][row] = sum_{o in head h} (scale Wk[o][c]) Q'[row][o]
#pragma unroll
        for (int h = 0; h < NT; ++h) {
#pragma unroll
            for (int P = 0; P < NP; ++P) {
                f32x4 acc[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int u = 2 * P + e;
                    const h16x4 wh = WkF[((h * NT + u) * 2) * 64 + lane], wl = WkF[((h * NT + u) * 2 + 1) * 64 + lane];
                    f32x4 mm = f32x4{0.f, 0.f, 0.f, 0.f}, cr = mm;
                    MFMA_H16(mm, wh, qh[h]);
                    MFMA_H16(cr, wh, ql[h]);
                    MFMA_H16(cr, wl, qh[h]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[e][i] = __builtin_fmaf(cr[i], H16_INV, mm[i]);
                }
                if (OUT == 1) {
                    h16x8 *dsth = reinterpret_cast<h16x8 *>(dst + h * CG);
                    h16x8 hi, lo;
                    h16_split8(acc[0], acc[1], hi, lo);
                    if (row_ok) {
                        dsth[(P * 4 + g) * 2] = hi;
                        dsth[(P * 4 + g) * 2 + 1] = lo;
                    }
                } else if (row_ok) {
                    store_handoff(dst + h * CG + (32 * P + 4 * g), acc[0]);
                    store_handoff(dst + h * CG + (32 * P + 16 + 4 * g), acc[1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef ATTN_Q16_ROWS
}

// ---- C, kv16 form: Xbar -> attention output rows
// VIN (round 6): the hand-off rows hold V (k_attn_kvh<.., VOUT>): only the output projection is left
template <int CG, int HP, bool VIN = false>
__global__ void __launch_bounds__(ATTN_QO16_WAVES *MSSVT_WAVE, 4) k_attn_o16(AttnPack pack) {
    const AttnArgs &a = pack.g[blockIdx.y];
    using L = AttnBlob<CG>;
    constexpr int NT = CG / 16, NP = CG / 32, QROW = HP * CG;
    extern __shared__ float4 lds4[];
    const h16x8 *WvF = reinterpret_cast<const h16x8 *>(lds4);
    const h16x8 *WoF = reinterpret_cast<const h16x8 *>(reinterpret_cast<const char *>(lds4) + (L::WO - L::WV));
    float *bv_l = reinterpret_cast<float *>(reinterpret_cast<char *>(lds4) + (L::WK2 - L::WV)), *bo_l = bv_l + CG;
    const int lane = lane_id(), r = lane & 15, g = lane >> 4;
    const int rows = *a.num_rows, tiles = (rows + 15) >> 4;
    const int wv = threadIdx.x / MSSVT_WAVE;
    if ((int)blockIdx.x >= tiles) return;
    int tile = wv * gridDim.x + blockIdx.x;
    f32x4 x[NT], xn[NT];  // piece S of one head's Xbar row: channels 16 S + 4 g + i
#define ATTN_O16_ROWS(dst_, row_, h_)                                                                     \
    {                                                                                                     \
        const float *xb_ = a.qbuf + (size_t)(row_) * QROW + (h_) * CG + 4 * g;                            \
        _Pragma("unroll") for (int S = 0; S < NT; ++S) {                                                  \
            const float4 t4_ = *reinterpret_cast<const float4 *>(xb_ + 16 * S);                           \
            dst_[S] = f32x4{t4_.x, t4_.y, t4_.z, t4_.w};                                                  \
        }                                                                                                 \
    }
    if (!VIN) ATTN_O16_ROWS(x, min(tile * 16 + r, rows - 1), 0)
    attn_stage16<(L::WK2 - L::WV) / 16, ATTN_QO16_WAVES * MSSVT_WAVE>(
        lds4, reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.packed) + L::WV));
    if (threadIdx.x < CG) {
        bv_l[threadIdx.x] = a.bkv[CG + threadIdx.x];
        bo_l[threadIdx.x] = a.bo[threadIdx.x];
    }
    __syncthreads();
    for (bool first = true; tile < tiles; tile += gridDim.x * ATTN_QO16_WAVES, first = false) {
        const int row = min(tile * 16 + r, rows - 1);
        const bool row_ok = tile * 16 + r < rows;
        const int dest = a.qrow_src[row].y;
        if (!first && !VIN) ATTN_O16_ROWS(x, row, 0)
        // GEMM3^T: V^T[o][row] = sum_c Wv[o][c] Xbar_{head(o)}[row][c] + bv[o]  (head of tile t = t)
        f32x4 v[NT];
        if (VIN) {  // V of head t: 16 floats at the start of the (row, head) slot, this lane's o = 16 t + 4 g + i
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 t4 = *reinterpret_cast<const float4 *>(a.qbuf + (size_t)row * QROW + t * CG + 4 * g);
                v[t] = f32x4{t4.x, t4.y, t4.z, t4.w};
            }
        }
#pragma unroll
        for (int t = 0; t < NT && !VIN; ++t) {
            if (t + 1 < NT) ATTN_O16_ROWS(xn, row, t + 1)  // the next head's row pieces under this head's products
            const float4 b = *reinterpret_cast<const float4 *>(bv_l + 16 * t + 4 * g);
            f32x4 mm = f32x4{b.x, b.y, b.z, b.w}, cr = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int P = 0; P < NP; ++P) {
                h16x8 xh, xl;
                h16_split8(x[2 * P], x[2 * P + 1], xh, xl);
                const h16x8 wh = WvF[((t * NP + P) * 2) * 64 + lane], wl = WvF[((t * NP + P) * 2 + 1) * 64 + lane];
                MFMA_H(mm, wh, xh);
                MFMA_H(cr, wh, xl);
                MFMA_H(cr, wl, xh);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[t][i] = __builtin_fmaf(cr[i], H16_INV, mm[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int S = 0; S < NT; ++S) x[S] = xn[S];
        }
        // GEMM4^T: out^T[p][row] = sum_o Wo[p][o] V[row][o] + bo[p]; two V tiles = one 32-wide step of the B operand
        h16x8 vh[NP], vl[NP];
#pragma unroll
        for (int s = 0; s < NP; ++s) h16_split8(v[2 * s], v[2 * s + 1], vh[s], vl[s]);
        float *dst = a.attn + (size_t)dest * a.C + a.c0 + 4 * g;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float4 b = *reinterpret_cast<const float4 *>(bo_l + 16 * u + 4 * g);
            f32x4 mm = f32x4{b.x, b.y, b.z, b.w}, cr = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const h16x8 wh = WoF[((u * NP + s) * 2) * 64 + lane], wl = WoF[((u * NP + s) * 2 + 1) * 64 + lane];
                MFMA_H(mm, wh, vh[s]);
                MFMA_H(cr, wh, vl[s]);
                MFMA_H(cr, wl, vh[s]);
            }
            if (row_ok)
                *reinterpret_cast<float4 *>(dst + 16 * u) = make_float4(__builtin_fmaf(cr[0], H16_INV, mm[0]), __builtin_fmaf(cr[1], H16_INV, mm[1]),
                                                                        __builtin_fmaf(cr[2], H16_INV, mm[2]), __builtin_fmaf(cr[3], H16_INV, mm[3]));
            if (u & 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef ATTN_O16_ROWS
}

// ---- B: keys, scores, softmax, xbar (one wavefront per window) ------------------------------------
// Per window the key tokens T[key][c] = xhat row + relu(pos. MLP) are needed in BOTH matrix-core
// operand layouts (lane l: a = l % 16, g = l / 16):
//   T1: lane (a = key % 16, g)  channels 16 S + 4 g + j   -> A operand of  S = T Qt^T   (reduce over c)
//   T2: lane (a = c % 16,  g)   keys     16 t + 4 g + i   -> A operand of  Xbar^T = T^T P (reduce over keys)
// T1 is gathered from HBM as 16-byte pieces of the rows and kept in registers; T2 is T1 transposed
// through a per-wave LDS tile (row stride Cg + 4: the b128 writes and the b32 column reads are both
// conflict free).  The positional MLP is a K = 4 product (rel.x, rel.y, rel.z, 1) x (wp0, wp1, wp2,
// window part) = one MFMA per 16 x 16 tile.  Queries go through in passes of 16 / HP (column n =
// query * HP + head); the softmax over keys is 8 in-lane values + two cross-row shuffles, and the
// normalised P accumulator IS the B operand of the second product.  Key slots are not compacted:
// masked slots score -inf (the reference adds -100: weight <= e^-100); key tiles without an unmasked
// slot are skipped.  Software pipeline across windows: metadata two windows ahead, the feature rows
// of the next window in flight under this window's MFMAs.
#ifdef MSSVT_STAMPS
// per wave: [0] entry, [1] first rows issued, [2] exit, [3] windows done, [4..9] cycles summed over its windows:
// token build (waits for the row gathers), prefetch issue, tile -> LDS, scores + softmax (waits for Qt), PV + store, passes
__device__ unsigned long long g_attn_kv_stamps[8192 * 12];
extern "C" int mssvt_debug_read_attn_kv_stamps(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_kv_stamps), sizeof(g_attn_kv_stamps));
}
#define KVS_T() __builtin_readcyclecounter()
#else
#define KVS_T() 0ull
#endif
template <int CG, int HD, int HP, int KT>
__global__ void __launch_bounds__(ATTN_ROW_WAVES *MSSVT_WAVE, 2) k_attn_kv(AttnPack pack) {
    const AttnArgs &a = pack.g[blockIdx.y];
    unsigned long long ks_entry = KVS_T(), ks_first = 0, ks_sum[6] = {0, 0, 0, 0, 0, 0}, ks_n = 0, ks_t = 0;
    (void)ks_entry; (void)ks_first; (void)ks_sum; (void)ks_n; (void)ks_t;
    constexpr int CGP = (CG + 15) / 16 * 16, NT = CGP / 16, NH = CG / HD, QROW = HP * CG, QPP = 16 / HP;
    constexpr int LSK = CGP + 4;  // LDS row stride of the key tile
    extern __shared__ float4 lds4[];
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = threadIdx.x / MSSVT_WAVE;
    float *Tl = reinterpret_cast<float *>(lds4) + (size_t)wv * KT * 16 * LSK;
    // positional MLP operand of this lane (channel 16 u + la, input g): constant part + window part
    float wconst[NT], w3[NT], w4[NT], w5[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int c = 16 * u + la;
        const bool in = CGP == CG || c < CG;
        const float *wp = a.Wp + (size_t)(a.c0 + (in ? c : 0)) * 6;
        wconst[u] = in ? (g < 3 ? wp[g] : a.bp[a.c0 + c]) : 0.f;
        w3[u] = in && g == 3 ? wp[3] : 0.f;
        w4[u] = in && g == 3 ? wp[4] : 0.f;
        w5[u] = in && g == 3 ? wp[5] : 0.f;
    }
    // static round-robin over the heaviest-first work order: wave i takes entries i, i + T, i + 2T ...
    // (T = waves in the grid), i.e. one window of every weight tier -- as balanced as dynamic tickets
    // without their atomics (a drained single-address ticket costs ~11 ns chip-wide, x 8192 waves)
    const int n_act = __builtin_amdgcn_readfirstlane(*a.num_wins);
    const int wstep = gridDim.x * ATTN_ROW_WAVES;
    const int K = a.K;
    int wi = __builtin_amdgcn_readfirstlane(blockIdx.x * ATTN_ROW_WAVES + wv);  // wave-uniform: scalar metadata loads
    if (wi >= n_act) return;
    // gathers through buffer descriptors: 32-bit lane offsets instead of 64-bit address arithmetic per load
    const __amdgpu_buffer_rsrc_t xr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.xhat), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t km_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(a.kmeta), 0, -1, 0x00020000);
    const unsigned row_bytes = (unsigned)a.C * 4u, lane_off = ((unsigned)a.c0 + 4u * g) * 4u;
#define KV_ROW4(off_, S_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr_rs, (off_) + 64u * (S_), 0, 0))
    // Software pipeline over the wave's windows with every load of the loop body UNCONDITIONAL (indices clamped
    // to the last window; rows of empty slots read row 0): a guarded load makes the wait counts path dependent and
    // the compiler falls back to vmcnt(0), which drains the prefetches.  vmcnt retires in order -- a wait for a
    // load also waits for every load issued before it -- hence window ids three steps ahead (stage P, scalar),
    // metadata two (stage M: centre / counts scalar, key slots vector), raw rows one (stage R).
    int w_p;
    // stage M: metadata of a window (perm -> kmeta would otherwise be dependent round trips per window)
    float4 wc_m, km_m[KT];
    int nqv_m, qbase_m;
#define KV_LOAD_META()                                                                     \
    {                                                                                      \
        wc_m = a.wcentre[w_p];                                                             \
        nqv_m = a.nq_valid[w_p];                                                           \
        qbase_m = a.q_off[w_p];                                                            \
        _Pragma("unroll") for (int t = 0; t < KT; ++t)                                     \
            km_m[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(    \
                km_rs, ((unsigned)w_p * (unsigned)K + (unsigned)min(16 * t + la, K - 1)) * 16u, 0, 0)); \
    }
    // stage R: resolved metadata + raw feature rows of a window (T1 layout)
    float4 wc_r;
    int nqv_r, qbase_r;
    float rel_r[KT];
    unsigned vmask_r;    // bit 4 t + i: key 16 t + 4 g + i is unmasked
    unsigned used_r;     // bit t: tile t has an unmasked key (wave-uniform)
    f32x4 T1n[KT][NT];
#define KV_ISSUE_ROWS()                                                                    \
    {                                                                                      \
        wc_r = wc_m; nqv_r = nqv_m; qbase_r = qbase_m;                                     \
        vmask_r = 0; used_r = 0;                                                           \
        _Pragma("unroll") for (int t = 0; t < KT; ++t) {                                   \
            const int r_ = __builtin_bit_cast(int, km_m[t].w);                             \
            const bool ok_ = 16 * t + la < K && r_ >= 0;                                   \
            const unsigned long long bal_ = __ballot(ok_);  /* bits 0..15: keys 16 t + 0..15 */ \
            vmask_r |= (unsigned)((bal_ >> (4 * g)) & 15ull) << (4 * t);                   \
            used_r |= (t == 0 || (bal_ & 0xFFFFull) != 0ull) ? 1u << t : 0u;               \
            rel_r[t] = lane_pick4(g, km_m[t].x, km_m[t].y, km_m[t].z, 1.0f); \
            const unsigned ro_ = (unsigned)__umul24((unsigned)(ok_ ? r_ : 0), row_bytes) + lane_off; \
            _Pragma("unroll") for (int S = 0; S < NT; ++S)                                 \
                T1n[t][S] = (CGP == CG || 16 * S + 4 * g < CG) ? KV_ROW4(ro_, S) : f32x4{0.f, 0.f, 0.f, 0.f}; \
        }                                                                                  \
    }
    const int w_last = n_act - 1;
    w_p = a.perm[wi];
    KV_LOAD_META()
    w_p = a.perm[min(wi + wstep, w_last)];
    KV_ISSUE_ROWS()
    KV_LOAD_META()
    w_p = a.perm[min(wi + 2 * wstep, w_last)];
    ks_first = KVS_T();
    for (; wi < n_act; wi += wstep) {
        ks_t = KVS_T();
        // ---- this window: stage R -> working registers ------------------------------------------------
        const float4 wc = wc_r;
        // a window whose rows would not fit the compact arrays is skipped (cannot happen with the caller's bound)
        const int nqv = qbase_r + nqv_r <= a.row_capacity ? nqv_r : 0;
        const size_t qbase = (size_t)qbase_r;
        const unsigned vmask = vmask_r, used = used_r;
        f32x4 T1[KT][NT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            if (!(used >> t & 1)) continue;
            // + relu(positional MLP): rows = channels 16 u + 4 g + i, column = key 16 t + la
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const float wu = ((wconst[u] + w3[u] * wc.x) + w4[u] * wc.y) + w5[u] * wc.z;
                f32x4 p1 = f32x4{0.f, 0.f, 0.f, 0.f};
                MFMA4(p1, wu, rel_r[t]);
#pragma unroll
                for (int i = 0; i < 4; ++i) T1[t][u][i] = T1n[t][u][i] + fmaxf(p1[i], 0.0f);
            }
        }
#ifdef MSSVT_STAMPS
        asm volatile("" :: "v"(T1[0][0][0]));
        { const unsigned long long t_ = KVS_T(); ks_sum[0] += t_ - ks_t; ks_t = t_; }
#endif
        // first query pass: its Qt rows travel while the key tile is transposed
        const int hh = la % HP;
        const bool head_ok = hh < NH;
        float *qrow = a.qbuf + (qbase + min(la / HP, nqv - 1)) * QROW + (head_ok ? hh : 0) * CG;
        f32x4 qt[NT];
#pragma unroll
        for (int S = 0; S < NT; ++S) {
            const int c = 16 * S + 4 * g;
            const float4 v = (CGP == CG || c < CG) ? *reinterpret_cast<const float4 *>(qrow + c)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            qt[S] = f32x4{v.x, v.y, v.z, v.w};
        }
        // ---- next window: rows in flight under this window's MFMAs, metadata one further ahead (past the end
        // of the work list the last window is fetched again) ------------------------------------------------
        KV_ISSUE_ROWS()
        KV_LOAD_META()
        w_p = a.perm[min(wi + 3 * wstep, w_last)];
#ifdef MSSVT_STAMPS
        { const unsigned long long t_ = KVS_T(); ks_sum[1] += t_ - ks_t; ks_t = t_; }
#endif
        // key tile -> LDS (the T2 operand is read back column-wise)
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            if (!(used >> t & 1)) continue;
#pragma unroll
            for (int S = 0; S < NT; ++S)
                *reinterpret_cast<float4 *>(Tl + (16 * t + la) * LSK + 16 * S + 4 * g) =
                    make_float4(T1[t][S][0], T1[t][S][1], T1[t][S][2], T1[t][S][3]);
        }
        wave_lds_sync();
#ifdef MSSVT_STAMPS
        { const unsigned long long t_ = KVS_T(); ks_sum[2] += t_ - ks_t; ks_t = t_; }
#endif
        // queries, QPP per pass: column la = query * HP + head
        for (int q0 = 0; q0 < nqv; q0 += QPP) {
            ++ks_sum[5];
            const int q = q0 + la / HP;
            const bool q_ok = q < nqv && head_ok;
            if (q0 > 0) {
                qrow = a.qbuf + (qbase + min(q, nqv - 1)) * QROW + (head_ok ? hh : 0) * CG;
#pragma unroll
                for (int S = 0; S < NT; ++S) {
                    const int c = 16 * S + 4 * g;
                    const float4 v = (CGP == CG || c < CG) ? *reinterpret_cast<const float4 *>(qrow + c)
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
                    qt[S] = f32x4{v.x, v.y, v.z, v.w};
                }
            }
            // scores: S[key][col] = sum_c T[key][c] Qt[col][c]; even / odd channel tiles accumulate
            // separately (two independent MFMA chains per key tile)
            f32x4 sc[KT], sc2[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) sc[t] = sc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            // Tiles in DESCENDING order, tile 0 (never empty: slot 0 of a list is never masked) unconditionally and
            // last: the reads below then follow a chain in straight-line code -- behind a branch that skips the tail
            // of the chain the compiler's padding came out one wait state short (tools/mfma_hazard_check.py)
#pragma unroll
            for (int t = KT - 1; t >= 0; --t) {
                if (t > 0 && !(used >> t & 1)) continue;
#pragma unroll
                for (int S = 0; S < NT; S += 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        MFMA4(sc[t], T1[t][S][j], qt[S][j]);
                        if (S + 1 < NT) MFMA4(sc2[t], T1[t][S + 1][j], qt[S + 1][j]);
                    }
                }
            }
            // softmax over the unmasked keys: lane (col, g) holds keys 16 t + 4 g + i
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    sc[t][i] += sc2[t][i];
                    mx = fmaxf(mx, (vmask >> (4 * t + i) & 1) ? sc[t][i] : -INFINITY);
                }
            mx = fmaxf(mx, lane_xor16(mx));
            mx = fmaxf(mx, lane_xor32(mx));
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = (vmask >> (4 * t + i) & 1) ? __expf(sc[t][i] - mx) : 0.0f;
                    sc[t][i] = e;
                    sum += e;
                }
            sum += lane_xor16(sum);
            sum += lane_xor32(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);  // slot 0 of a list is never masked: sum >= 1
#ifdef MSSVT_STAMPS
            asm volatile("" :: "v"(inv));
            { const unsigned long long t_ = KVS_T(); ks_sum[3] += t_ - ks_t; ks_t = t_; }
#endif
            // Xbar^T[c][col] = sum_key T[key][c] P[key][col]; A operand = the LDS tile read column-wise
            f32x4 acc[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = KT - 1; t >= 0; --t) {  // (descending, tile 0 unconditionally: as for the scores)
                if (t > 0 && !(used >> t & 1)) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pv = sc[t][i] * inv;
                    const float *col = Tl + (16 * t + 4 * g + i) * LSK + la;
                    float tv[NT];
#pragma unroll
                    for (int u = 0; u < NT; ++u) tv[u] = col[16 * u];
#pragma unroll
                    for (int u = 0; u < NT; ++u) MFMA4(acc[u], tv[u], pv);
                }
            }
            if (q_ok) {
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const int c = 16 * u + 4 * g;
                    if (CGP == CG || c < CG)  // xbar replaces qt in place (this lane's own 16 bytes)
                        store_handoff(qrow + c, acc[u]);
                }
            }
#ifdef MSSVT_STAMPS
            { const unsigned long long t_ = KVS_T(); ks_sum[4] += t_ - ks_t; ks_t = t_; }
#endif
        }
        wave_lds_sync();  // the next window rewrites the tile
        ++ks_n;
    }
#ifdef MSSVT_STAMPS
    {
        const int wid = (blockIdx.y * gridDim.x + blockIdx.x) * ATTN_ROW_WAVES + wv;
        if (lane == 0 && wid < 8192) {
            unsigned long long *o = g_attn_kv_stamps + (size_t)wid * 12;
            o[0] = ks_entry; o[1] = ks_first; o[2] = KVS_T(); o[3] = ks_n;
            for (int i = 0; i < 6; ++i) o[4 + i] = ks_sum[i];
        }
    }
#endif
#undef ATTN_Q_ROWS
#undef ATTN_O_ROWS
#undef KV_LOAD_META
#undef KV_ISSUE_ROWS
#undef KV_ROW4
}


// ---- B, kv16 form: the same launch with split-fp16 matrix operands --------------------------------------------
// Launch B above is bound by the fp32 matrix instruction (v_mfma_f32_16x16x4_f32, 32 cycles for a 16 x 16 x 4 tile: 72 of
// them per window, pass and head group at K = 32 -- ~2.3 k of the ~3.2 k cycles a window keeps its SIMD busy).  Here every
// product sum is three v_mfma_f32_16x16x32_f16 on (hi, lo) halves (see h16_split4; fp32 accumulation, the error of the fp32
// instruction as in csrc/ffn.hip): 12 + 12 sixteen-cycle instructions per window and pass instead of 32 + 32 thirty-two-cycle
// ones.  What makes it cheaper here than in a single-launch form that projects the keys (round 2: ~20 fragment splits per window, slower): the
// key tokens are the only operand split in this kernel, ONCE per window (Qt arrives split from k_attn_q<KV16>, P is 8
// values per lane and pass), and the transposed operand of the second product comes out of the LDS image by
// ds_read_b64_tr_b16 instead of 32 scalar column reads.
//   channels: k slot (g, j) of step P <-> channel 32 P + 16 (j / 4) + 4 g + j % 4 in BOTH operands of the score product (two
//             dense 16-byte pieces of a key row per step; Qt is stored in that order by launch A); the image keeps the
//             lane's 8 slots together (column 32 P + 8 g + j), so the second product's output rows come out in slot
//             order and the Xbar store undoes the permutation in its address;
//   keys:     k slot (g, j) of step s <-> key 32 s + 16 (j / 4) + 4 g + j % 4 = the accumulator layout of two score tiles,
//             so the normalised scores are the B operand of the second product as they stand.
// LDS image per wave: [hi | lo][key][channel] fp16, rows of 2 CG + 16 bytes (36 KiB per workgroup at K = 32: four
// workgroups = 4 waves / SIMD per CU; the 8 rows a 32-lane half of a transposed read touches start 36 banks apart).  Unused key tiles hold zeros (their P is 0, but the product needs finite
// operands).  The caller guarantees the fp16 range of tokens and Qt (fused._attn_kv16_ok).
#define KVH_RS(cg) (2 * (cg) + 16)  // bytes per image row: 16-byte aligned, 36 banks at Cg = 64 (transposed reads: one 2-way pair per half)
__device__ __forceinline__ h16x4 lds_read_tr16(const char *p) {
    return __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) fp16v4 *)(p)));
}
// KVH_QT_AHEAD = 1: the first-pass Qt fragments of a window are requested one window ahead like its key rows (16 more
// VGPRs: 3 waves / SIMD).  Measured on one box (tools/ab_kvh.sh): 94.2 / 46.9 us per Block attention against 91.6 / 45.5
// without -- the window launch is not waiting for these loads, it is short of vector-memory issue slots (round-3 timing-only ablation builds, DESIGN.md section 4:
// dropping the 4 Qt loads of a pass -7.3 us, the 4 Xbar stores -5.9 us, serving the key rows from 8 hot rows -0.6 us).
#ifndef KVH_QT_AHEAD
#define KVH_QT_AHEAD 0
#endif
// QP: the hand-off rows hold Q' fragments (k_attn_q16<.., 2>) and Qt_h = (scale Wk_h)^T q'_h is formed here, per pass, from the
// Wk fragments of the pack blob staged into the LDS: one product per head with the columns of the other heads zeroed in
// the B operand, all accumulated into one tile -- NH x NT x 3 K = 16 instructions for 1 instead of 4 row loads per lane
// VOUT (round 6, MSSVT_ATTN_VFUSE=1): V_h = Wv_h Xbar_h + bv_h is formed HERE, per pass, and the hand-off row holds V (16 floats per
// query and head) instead of Xbar (Cg floats): every output tile of Wv (= one head, HD = 16) times the pass's Xbar columns, a
// lane keeps the tile of ITS column's head -- NT x NP x 3 more instructions and one operand split per pass for a quarter of
// the hand-off bytes; k_attn_o16<.., VIN> then only applies Wo.  The Wv fragments (16 KiB per group) join the Wk pair
// fragments in the LDS, so the workgroup becomes NWV = 12 waves sharing one copy: 32 + 12 x 9 KiB, still 3 waves per SIMD.
template <int CG, int HD, int HP, int KT, bool QP, bool VOUT = false, int NWV = ATTN_ROW_WAVES>
__global__ void __launch_bounds__(NWV *MSSVT_WAVE, KT <= 2 ? (KVH_QT_AHEAD || QP ? 3 : 4) : 2) k_attn_kvh(AttnPack pack) {
    static_assert(!QP || (HD == 16 && (CG / HD) % 2 == 0 && !KVH_QT_AHEAD), "Q' hand-off: head = one 16-row tile, heads in pairs");
    static_assert(!VOUT || (QP && HP * HD == CG), "V hand-off: Q' mode, one output tile of Wv per head");
    static_assert(CG % 32 == 0 && KT % 2 == 0, "32-channel and 32-key steps");
    const AttnArgs &a = pack.g[blockIdx.y];
    constexpr int NT = CG / 16, NP = CG / 32, NS = KT / 2, NH = CG / HD, QROW = HP * CG, QPP = 16 / HP;
    constexpr int RS = KVH_RS(CG), IMG = KT * 16 * RS;  // bytes: image row, one (hi or lo) image
    extern __shared__ float4 lds4[];
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = threadIdx.x / MSSVT_WAVE;
    // bytes of fragments in front of the images: the Wk pair fragments (+ the Wv fragments of the V hand-off)
    constexpr int WKB = QP ? (VOUT ? AttnBlob<CG>::BYTES : AttnBlob<CG>::WV2) - AttnBlob<CG>::WK2 : 0;
    const h16x8 *WvF2 = reinterpret_cast<const h16x8 *>(reinterpret_cast<const char *>(lds4) + (AttnBlob<CG>::WV2 - AttnBlob<CG>::WK2));
    (void)WvF2;
    char *Ti = reinterpret_cast<char *>(lds4) + WKB + (size_t)wv * 2 * IMG;
    const h16x8 *WkF2 = reinterpret_cast<const h16x8 *>(lds4);
    if (QP) {  // before any wave can leave: every wave of the workgroup meets at the barrier
        attn_stage16<(WKB > 0 ? WKB : 16) / 16, NWV * MSSVT_WAVE>(
            lds4, reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.packed) + AttnBlob<CG>::WK2));
        __syncthreads();
    }
    // positional MLP operand of this lane: A row la of tile u <-> channel (16 * u + la), input g: the weight of the
    // relative offset (g < 3) | bias + window-centre part (g = 3: summed over the three lanes that hold its weights)
    float wrel[NT], wctr[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int c = (16 * u + la);
        const float *wp = a.Wp + (size_t)(a.c0 + c) * 6;
        wrel[u] = g < 3 ? wp[g] : a.bp[a.c0 + c];
        wctr[u] = g < 3 ? wp[3 + g] : 0.f;
    }
    const int n_act = __builtin_amdgcn_readfirstlane(*a.num_wins);
    const int wstep = gridDim.x * NWV;
    const int K = a.K;
    // XCD-aware deal of the work order (MSSVT_XCD_REMAP=2; off by default until measured faster): neighbours in the order are
    // neighbours in space inside a weight class and share key rows, and every XCD has an L2 of its own.  Whole contiguous runs
    // per XCD (common.hip.h, xcd_contiguous_block) cut the HBM traffic 114.1 -> 106.3 MB per launch but cost 106k -> 116k cycles:
    // the order is heaviest-first, and a contiguous eighth of a round hands one XCD the heaviest windows of every round.  So:
    // CHUNKS of 24 consecutive entries, dealt round-robin to the XCDs -- every XCD sees the whole weight range of a round.
    int wi;
    {
        const int per_xcd = (int)(gridDim.x >> 3) * NWV;
        if (a.xcd > 1 && (gridDim.x & 7) == 0 && per_xcd % 24 == 0) {
            const int x = blockIdx.x & 7, idx = (int)(blockIdx.x >> 3) * NWV + wv;
            wi = __builtin_amdgcn_readfirstlane(((idx / 24) * 8 + x) * 24 + idx % 24);
        } else {
            wi = __builtin_amdgcn_readfirstlane(blockIdx.x * NWV + wv);
        }
    }
    if (wi >= n_act) return;
    const __amdgpu_buffer_rsrc_t xr_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.xhat), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t km_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(a.kmeta), 0, -1, 0x00020000);
    const unsigned row_bytes = (unsigned)a.C * 4u, lane_off = ((unsigned)a.c0 + 4u * g) * 4u;
    // piece S of a row: channels 16 S + 4 g + i (dense: the four g lanes of a key read 64 contiguous bytes)
#define KVH_ROW4(off_, S_) __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr_rs, (off_) + 64u * (S_), 0, 0))
    // the software pipeline of k_attn_kv: window ids three steps ahead, metadata two, raw rows one; every load unconditional
    int w_p;
    float4 wc_m, km_m[KT];
    int nqv_m, qbase_m;
#define KVH_LOAD_META()                                                                    \
    {                                                                                      \
        wc_m = a.wcentre[w_p];                                                             \
        nqv_m = a.nq_valid[w_p];                                                           \
        qbase_m = a.q_off[w_p];                                                            \
        _Pragma("unroll") for (int t = 0; t < KT; ++t)                                     \
            km_m[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(    \
                km_rs, ((unsigned)w_p * (unsigned)K + (unsigned)min(16 * t + la, K - 1)) * 16u, 0, 0)); \
    }
    float4 wc_r;
    int nqv_r, qbase_r;
    float rel_r[KT];
    unsigned vmask_r, used_r;
    f32x4 T1n[KT][NT];
    // (round 6 measured the first pass's Q' piece travelling with the rows, one window ahead -- the load that cost the CEILING
    // kernel 6 us per odd launch -- in this kernel: 52.1 / 26.0 against 51.4 / 25.6 us, no gain: three waves per SIMD hide it)
#define KVH_ISSUE_ROWS()                                                                   \
    {                                                                                      \
        wc_r = wc_m; nqv_r = nqv_m; qbase_r = qbase_m;                                     \
        vmask_r = 0; used_r = 0;                                                           \
        _Pragma("unroll") for (int t = 0; t < KT; ++t) {                                   \
            const int r_ = __builtin_bit_cast(int, km_m[t].w);                             \
            const bool ok_ = 16 * t + la < K && r_ >= 0;                                   \
            const unsigned long long bal_ = __ballot(ok_);                                 \
            vmask_r |= (unsigned)((bal_ >> (4 * g)) & 15ull) << (4 * t);                   \
            used_r |= (t == 0 || (bal_ & 0xFFFFull) != 0ull) ? 1u << t : 0u;               \
            rel_r[t] = lane_pick4(g, km_m[t].x, km_m[t].y, km_m[t].z, 1.0f); \
            const unsigned ro_ = (unsigned)__umul24((unsigned)(ok_ ? r_ : 0), row_bytes) + lane_off; \
            _Pragma("unroll") for (int S = 0; S < NT; ++S) T1n[t][S] = KVH_ROW4(ro_, S);   \
        }                                                                                  \
    }
    const int w_last = n_act - 1;
    w_p = a.perm[wi];
    KVH_LOAD_META()
    w_p = a.perm[min(wi + wstep, w_last)];
    const int hh = la % HP;
    const bool head_ok = hh < NH;
    // V hand-off: this lane's four outputs of its column's head, o = 16 hh + 4 g + i
    const float4 bv4 = VOUT ? *reinterpret_cast<const float4 *>(a.bkv + CG + 16 * (head_ok ? hh : 0) + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    // first-pass Qt fragments of a window travel one window ahead as well (stage R): issued before the previous window's
    // Xbar stores, so that waiting for them never waits for those stores (vmcnt retires in order)
    h16x8 qh_r[NP], ql_r[NP];
    (void)qh_r; (void)ql_r;
#define KVH_ISSUE_QT()                                                                     \
    {                                                                                      \
        const int nq_ = qbase_r + nqv_r <= a.row_capacity ? nqv_r : 0;                     \
        const h16x8 *qr_ = reinterpret_cast<const h16x8 *>(                                \
            a.qbuf + ((size_t)qbase_r + max(min(la / HP, nq_ - 1), 0)) * QROW + (head_ok ? hh : 0) * CG); \
        _Pragma("unroll") for (int P = 0; P < NP; ++P) {                                   \
            qh_r[P] = qr_[(P * 4 + g) * 2];                                                \
            ql_r[P] = qr_[(P * 4 + g) * 2 + 1];                                            \
        }                                                                                  \
    }
    KVH_ISSUE_ROWS()
#if KVH_QT_AHEAD
    KVH_ISSUE_QT()
#endif
    KVH_LOAD_META()
    w_p = a.perm[min(wi + 2 * wstep, w_last)];
    // transposed reads: lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 of its 4 x 16 block
    const char *tr_base = Ti + (4 * g + (la >> 2)) * RS + 8 * (la & 3);
    for (; wi < n_act; wi += wstep) {
        const float4 wc = wc_r;
        const int nqv = qbase_r + nqv_r <= a.row_capacity ? nqv_r : 0;
        const size_t qbase = (size_t)qbase_r;
        const unsigned vmask = vmask_r, used = used_r;
        const float ctrb = lane_pick4(g, wc.x, wc.y, wc.z, wc.z);  // wctr is 0 for g = 3
        // key tokens = row + relu(positional MLP), split once, straight into the image (both products read it: the score
        // product row-wise, the second one transposed -- no token registers live across the passes: 4 waves / SIMD)
        float wu[NT];  // A operand of the positional product: once per window (the window centre), not once per key tile
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            float cs = wctr[u] * ctrb;
            cs += lane_xor16(cs);
            cs += lane_xor32(cs);
            wu[u] = g == 3 ? wrel[u] + cs : wrel[u];
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) {
#pragma unroll
            for (int P = 0; P < NP; ++P) {
                h16x8 th = h16x8{0, 0, 0, 0, 0, 0, 0, 0}, tl = th;
                if (used >> t & 1) {
                    f32x4 tk[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int u = 2 * P + h;
                        f32x4 p1 = f32x4{0.f, 0.f, 0.f, 0.f};
                        MFMA4(p1, wu[u], rel_r[t]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) tk[h][i] = T1n[t][u][i] + fmaxf(p1[i], 0.0f);
                    }
                    h16_split8(tk[0], tk[1], th, tl);
                }
                char *dst = Ti + (16 * t + la) * RS + 64 * P + 16 * g;  // key row 16 t + la, the lane's 8 k slots of step P
                *reinterpret_cast<h16x8 *>(dst) = th;
                *reinterpret_cast<h16x8 *>(dst + IMG) = tl;
            }
        }
        h16x8 qh[NP], ql[NP];
#if KVH_QT_AHEAD
#pragma unroll
        for (int P = 0; P < NP; ++P) {
            qh[P] = qh_r[P];
            ql[P] = ql_r[P];
        }
#else
        h16x8 qp8 = h16x8{0, 0, 0, 0, 0, 0, 0, 0};  // Q' mode: (hi x 4 | lo x 4) of this lane's column
        if (QP) {
            qp8 = reinterpret_cast<const h16x8 *>(a.qbuf + (qbase + max(min(la / HP, nqv - 1), 0)) * QROW)[(head_ok ? hh : 0) * 4 + g];
        } else {
            const h16x8 *qr_ = reinterpret_cast<const h16x8 *>(a.qbuf + (qbase + max(min(la / HP, nqv - 1), 0)) * QROW + (head_ok ? hh : 0) * CG);
#pragma unroll
            for (int P = 0; P < NP; ++P) {
                qh[P] = qr_[(P * 4 + g) * 2];
                ql[P] = qr_[(P * 4 + g) * 2 + 1];
            }
        }
#endif
        KVH_ISSUE_ROWS()
#if KVH_QT_AHEAD
        KVH_ISSUE_QT()
#endif
        KVH_LOAD_META()
        w_p = a.perm[min(wi + 3 * wstep, w_last)];
        wave_lds_sync();
        for (int q0 = 0; q0 < nqv; q0 += QPP) {
            const int q = q0 + la / HP;
            const bool q_ok = q < nqv && head_ok;
            float *xrow = a.qbuf + (qbase + min(q, nqv - 1)) * QROW + (head_ok ? hh : 0) * CG;
            if (q0 > 0) {
                if (QP) {
                    qp8 = reinterpret_cast<const h16x8 *>(a.qbuf + (qbase + min(q, nqv - 1)) * QROW)[(head_ok ? hh : 0) * 4 + g];
                } else {
                    const h16x8 *qrow = reinterpret_cast<const h16x8 *>(xrow);
#pragma unroll
                    for (int P = 0; P < NP; ++P) {
                        qh[P] = qrow[(P * 4 + g) * 2];
                        ql[P] = qrow[(P * 4 + g) * 2 + 1];
                    }
                }
            }
            if (QP) {
                // Qt^T[c][col] = sum_h sum_{k < 16} (scale Wk)[16 h + k][c] Q'[q(col)][16 h + k] [h == head(col)]
                const h16x4 z4 = h16x4{0, 0, 0, 0};
                const h16x4 bh = h16x4{qp8[0], qp8[1], qp8[2], qp8[3]}, bl = h16x4{qp8[4], qp8[5], qp8[6], qp8[7]};
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler would hoist all 32 fragment reads: 70 spilled registers)
#pragma unroll
                for (int P = 0; P < NP; ++P) {
                    f32x4 qt[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int u = 2 * P + e;
                        f32x4 mm = f32x4{0.f, 0.f, 0.f, 0.f}, cr = mm;
#pragma unroll
                        for (int pr = 0; pr < NH / 2; ++pr) {  // heads 2 pr | 2 pr + 1 in the two halves of the k slots
                            const h16x8 wh = WkF2[((pr * NT + u) * 2) * 64 + lane], wl = WkF2[((pr * NT + u) * 2 + 1) * 64 + lane];
                            const h16x8 sh = h16_cat(hh == 2 * pr ? bh : z4, hh == 2 * pr + 1 ? bh : z4),
                                        sl = h16_cat(hh == 2 * pr ? bl : z4, hh == 2 * pr + 1 ? bl : z4);
                            MFMA_H(mm, wh, sh);
                            MFMA_H(cr, wh, sl);
                            MFMA_H(cr, wl, sh);
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) qt[e][i] = __builtin_fmaf(cr[i], H16_INV, mm[i]);
                    }
                    h16_split8(qt[0], qt[1], qh[P], ql[P]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // scores S[key][col] = sum_c T[key][c] Qt[col][c]
            f32x4 sc[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!(used >> t & 1)) continue;
                f32x4 mm = sc[t], cr = sc[t];
#pragma unroll
                for (int P = 0; P < NP; ++P) {
                    const char *src = Ti + (16 * t + la) * RS + 64 * P + 16 * g;
                    const h16x8 th = *reinterpret_cast<const h16x8 *>(src), tl = *reinterpret_cast<const h16x8 *>(src + IMG);
                    MFMA_H(mm, th, qh[P]);
                    MFMA_H(cr, th, ql[P]);
                    MFMA_H(cr, tl, qh[P]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) sc[t][i] = __builtin_fmaf(cr[i], H16_INV, mm[i]);
            }
            // softmax over the unmasked keys in base 2 (QP: log2 e is folded into the Wk fragments): masked slots score -inf
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sv = QP ? sc[t][i] : sc[t][i] * 1.4426950408889634f;
                    sc[t][i] = (vmask >> (4 * t + i) & 1) ? sv : -INFINITY;
                    mx = fmaxf(mx, sc[t][i]);
                }
            mx = fmaxf(mx, lane_xor16(mx));
            mx = fmaxf(mx, lane_xor32(mx));
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = __builtin_amdgcn_exp2f(sc[t][i] - mx);  // slot 0 of a list is never masked: mx is finite
                    sc[t][i] = e;
                    sum += e;
                }
            sum += lane_xor16(sum);
            sum += lane_xor32(sum);
            const float inv = __builtin_amdgcn_rcpf(sum);
            // Xbar^T[c][col] = sum_key T[key][c] P[key][col]
            h16x8 ph[NS], pl[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) h16_split8(sc[2 * s] * inv, sc[2 * s + 1] * inv, ph[s], pl[s]);
            f32x4 xb[VOUT ? NT : 1];  // V hand-off: the pass's Xbar tiles stay in registers
            (void)xb;
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                f32x4 mm = f32x4{0.f, 0.f, 0.f, 0.f}, cr = mm;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const char *blk = tr_base + 32 * s * RS + 32 * u;
                    const h16x8 ah = h16_cat(lds_read_tr16(blk), lds_read_tr16(blk + 16 * RS));
                    const h16x8 al = h16_cat(lds_read_tr16(blk + IMG), lds_read_tr16(blk + IMG + 16 * RS));
                    MFMA_H(mm, ah, ph[s]);
                    MFMA_H(cr, ah, pl[s]);
                    MFMA_H(cr, al, ph[s]);
                }
                const f32x4 xt = f32x4{__builtin_fmaf(cr[0], H16_INV, mm[0]), __builtin_fmaf(cr[1], H16_INV, mm[1]),
                                       __builtin_fmaf(cr[2], H16_INV, mm[2]), __builtin_fmaf(cr[3], H16_INV, mm[3])};
                if (VOUT) {
                    xb[VOUT ? u : 0] = xt;
                } else if (q_ok) {  // xbar replaces qt in place (this lane's own bytes of the row)
                    // image column 16 u + 4 g + i is k slot (2 (u % 2) + g / 2, 4 (g % 2) + i) of step u / 2 (see above)
                    store_handoff(xrow + 32 * (u >> 1) + 16 * (g & 1) + 8 * (u & 1) + 4 * (g >> 1), xt);
                }
            }
            if (VOUT) {
                // V^T[o][col] = sum_c Wv[o][c] Xbar[c][col] for every head tile t; the lane keeps t = head(col).  Tiles 2 P,
                // 2 P + 1 of Xbar are the two halves of k step P as they stand (AttnBlob::WV2 holds Wv in that order)
                h16x8 xh[NP], xl[NP];
#pragma unroll
                for (int P = 0; P < NP; ++P) h16_split8(xb[VOUT ? 2 * P : 0], xb[VOUT ? 2 * P + 1 : 0], xh[P], xl[P]);
                f32x4 vsel = f32x4{0.f, 0.f, 0.f, 0.f};
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 mm = f32x4{0.f, 0.f, 0.f, 0.f}, cr = mm;
#pragma unroll
                    for (int P = 0; P < NP; ++P) {
                        const h16x8 wh = WvF2[((t * NP + P) * 2) * 64 + lane], wl = WvF2[((t * NP + P) * 2 + 1) * 64 + lane];
                        MFMA_H(mm, wh, xh[P]);
                        MFMA_H(cr, wh, xl[P]);
                        MFMA_H(cr, wl, xh[P]);
                    }
                    const bool mine = hh == t;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float vt = __builtin_fmaf(cr[i], H16_INV, mm[i]);
                        vsel[i] = mine ? vt : vsel[i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x4 vout = f32x4{vsel[0] + bv4.x, vsel[1] + bv4.y, vsel[2] + bv4.z, vsel[3] + bv4.w};
                if (q_ok) store_handoff(xrow + 4 * g, vout);  // V of (query, head): the first 16 floats of its slot
            }
        }
        wave_lds_sync();  // the next window rewrites the image
    }
#undef KVH_LOAD_META
#undef KVH_ISSUE_QT
#undef KVH_ISSUE_ROWS
#undef KVH_ROW4
}

template <int CG, int HD, int HP>
static int launch_block_attn(const AttnPack &pack, int ng, int row_capacity, bool kv16, hipStream_t stream) {
    constexpr int CGP = (CG + 15) / 16 * 16, LS = CGP + 4;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    // A / C: persistent over 16-row tiles, one 16-wave workgroup per CU and head group at most
    const int tiles_cap = (row_capacity + 15) / 16;
    int row_grid = tiles_cap;
    static const int qo_wgs = getenv("MSSVT_ATTN_QO_WGS") ? atoi(getenv("MSSVT_ATTN_QO_WGS")) : 2;
    if (row_grid > cus * qo_wgs / (2 * ng)) row_grid = cus * qo_wgs / (2 * ng);
    if (row_grid < 1) row_grid = 1;
    const size_t lds_q = ((size_t)2 * CGP * LS + CGP * 8 + CGP) * 4, lds_o = ((size_t)2 * CGP * LS + 2 * CGP) * 4;
    const int K = pack.g[0].K;
    if constexpr (CG % 32 == 0) {
        if (kv16 && K > 16 && K <= 64) {  // split-fp16 operands in launch B (k_attn_kvh); A writes Qt pre-split
            constexpr int RS = KVH_RS(CG);
            static const int kvh_wgs = getenv("MSSVT_ATTN_KVH_WGS") ? atoi(getenv("MSSVT_ATTN_KVH_WGS")) : 0;
            const int wgs = kvh_wgs > 0 ? kvh_wgs : (K <= 32 ? (KVH_QT_AHEAD ? 3 : 4) : 2);  // resident workgroups per CU
            const dim3 kv_grid(cus * wgs / ng > 0 ? cus * wgs / ng : 1, ng);
            const size_t img = (size_t)ATTN_ROW_WAVES * 2 * 16 * RS;  // per key tile of 16 slots, all waves, hi + lo
            bool packed = HD == 16 && !KVH_QT_AHEAD;
            for (int g = 0; g < ng; ++g) packed = packed && pack.g[g].packed != nullptr;
            int grid16 = tiles_cap;  // 8-wave workgroups, two per CU
            if (grid16 > cus * 2 / ng) grid16 = cus * 2 / ng;
            if (grid16 < 1) grid16 = 1;
            static const int qo16 = getenv("MSSVT_ATTN_QO16_MASK") ? atoi(getenv("MSSVT_ATTN_QO16_MASK")) : 7;  // 1: A, 2: C, 4: Q' hand-off
            const bool q16 = packed && (qo16 & 1), o16 = packed && (qo16 & 2), qp = q16 && (qo16 & 4) && K <= 32;
            static const int vfuse_env = getenv("MSSVT_ATTN_VFUSE") ? atoi(getenv("MSSVT_ATTN_VFUSE")) : 0;
            const bool vfuse = (vfuse_env == 1 || vfuse_env == 2) && qp && o16 && HP * HD == CG;  // V formed in the window launch (round 6)
            if constexpr (HD == 16 && !KVH_QT_AHEAD) {
                if (qp) {
                    using L = AttnBlob<CG>;
                    const dim3 qp_grid(cus * 3 / ng > 0 ? cus * 3 / ng : 1, ng);
                    k_attn_q16<CG, HP, 2><<<dim3(grid16, ng), ATTN_QO16_WAVES * MSSVT_WAVE, L::WK + CG * 4, stream>>>(pack);
                    if constexpr (HP * HD == CG) {
#define KVH_V_LAUNCH(VOUT_, NWV_)                                                                                              \
    {                                                                                                                          \
        const size_t lds_v = (size_t)((VOUT_ ? L::BYTES : L::WV2) - L::WK2) + (size_t)(NWV_) * 2 * 2 * 16 * RS;               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_attn_kvh<CG, HD, HP, 2, true, VOUT_, NWV_>),       \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v);                           \
        if (e != hipSuccess) return (int)e;                                                                                    \
        const int per_cu = 12 / (NWV_);                                                                                        \
        const dim3 v_grid(cus * per_cu / ng > 0 ? cus * per_cu / ng : 1, ng);                                                  \
        k_attn_kvh<CG, HD, HP, 2, true, VOUT_, NWV_><<<v_grid, (NWV_) * MSSVT_WAVE, lds_v, stream>>>(pack);                    \
    }
                        // V hand-off (experiments): 1 = one 12-wave workgroup per CU and group half (the
                        // fragments are shared), 2 = 4-wave workgroups (68 KB each: two per CU), 3 = X-bar hand-off on 12-wave workgroups
                        if (vfuse && vfuse_env == 1) KVH_V_LAUNCH(true, 12)
                        else if (vfuse && vfuse_env == 2) KVH_V_LAUNCH(true, 4)
                        else if (vfuse_env == 3) KVH_V_LAUNCH(false, 12)
#undef KVH_V_LAUNCH
                    }
                    if (!((vfuse || vfuse_env == 3) && HP * HD == CG))
                        k_attn_kvh<CG, HD, HP, 2, true><<<qp_grid, ATTN_ROW_WAVES * MSSVT_WAVE, (L::WV2 - L::WK2) + 2 * img, stream>>>(pack);
                }
            }
            if constexpr (HD == 16) {
                if (q16 && !qp)
                    k_attn_q16<CG, HP, 1><<<dim3(grid16, ng), ATTN_QO16_WAVES * MSSVT_WAVE, AttnBlob<CG>::WV + CG * 4, stream>>>(pack);
            }
            if (!q16)
                k_attn_q<CG, HD, HP, true><<<dim3(row_grid, ng), ATTN_QO_WAVES * MSSVT_WAVE, lds_q, stream>>>(pack);
            if (qp) {
            } else if (K <= 32)
                k_attn_kvh<CG, HD, HP, 2, false><<<kv_grid, ATTN_ROW_WAVES * MSSVT_WAVE, 2 * img, stream>>>(pack);
            else {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_attn_kvh<CG, HD, HP, 4, false>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * img));
                if (e != hipSuccess) return (int)e;
                k_attn_kvh<CG, HD, HP, 4, false><<<kv_grid, ATTN_ROW_WAVES * MSSVT_WAVE, 4 * img, stream>>>(pack);
            }
            if constexpr (HD == 16) {
                if (o16 && vfuse)
                    k_attn_o16<CG, HP, true><<<dim3(grid16, ng), ATTN_QO16_WAVES * MSSVT_WAVE, AttnBlob<CG>::WK2 - AttnBlob<CG>::WV + 2 * CG * 4, stream>>>(pack);
                else if (o16)
                    k_attn_o16<CG, HP><<<dim3(grid16, ng), ATTN_QO16_WAVES * MSSVT_WAVE, AttnBlob<CG>::WK2 - AttnBlob<CG>::WV + 2 * CG * 4, stream>>>(pack);
            }
            if (!o16)
                k_attn_o<CG, HD, HP><<<dim3(row_grid, ng), ATTN_QO_WAVES * MSSVT_WAVE, lds_o, stream>>>(pack);
            return mssvt_launch_status();
        }
    }
    k_attn_q<CG, HD, HP, false><<<dim3(row_grid, ng), ATTN_QO_WAVES * MSSVT_WAVE, lds_q, stream>>>(pack);
    // B: persistent over the work order with exactly the waves that are resident (3 workgroups of 4 waves per CU at the
    // 164 VGPRs of the K = 32 instantiation): every further workgroup would run in a later round and pay the prologue
    // (positional weights, three dependent metadata round trips: ~9 k cycles, as much as one window) again for its few
    // windows -- 8 per CU: 47.0 us mean per launch, 3 per CU: 42.9 us.  Measured and dropped (round 3): one wave per
    // window without a work list (the ~37 k empty workgroups of a capacity-sized grid cost 5 k cycles each: 92 us), and
    // the same pipeline squeezed under 128 VGPRs for 4 waves / SIMD (18 spilled registers, 63 us): per window the SIMD
    // is busy ~3.2 k of the wave's 9.7 k cycles, more than half of it fp32 matrix instructions -- not a latency problem.
    static const int kv_wgs = getenv("MSSVT_ATTN_KV_WGS") ? atoi(getenv("MSSVT_ATTN_KV_WGS")) : 3;
    const dim3 kv_grid(cus * kv_wgs / ng > 0 ? cus * kv_wgs / ng : 1, ng);
    const size_t lds_tile = (size_t)ATTN_ROW_WAVES * 16 * LS * 4;  // per key tile of 16 slots, all waves
    if (K <= 16)
        k_attn_kv<CG, HD, HP, 1><<<kv_grid, ATTN_ROW_WAVES * MSSVT_WAVE, lds_tile, stream>>>(pack);
    else if (K <= 32)
        k_attn_kv<CG, HD, HP, 2><<<kv_grid, ATTN_ROW_WAVES * MSSVT_WAVE, 2 * lds_tile, stream>>>(pack);
    else {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_attn_kv<CG, HD, HP, 4>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * lds_tile));
        if (e != hipSuccess) return (int)e;
        k_attn_kv<CG, HD, HP, 4><<<kv_grid, ATTN_ROW_WAVES * MSSVT_WAVE, 4 * lds_tile, stream>>>(pack);
    }
    k_attn_o<CG, HD, HP><<<dim3(row_grid, ng), ATTN_QO_WAVES * MSSVT_WAVE, lds_o, stream>>>(pack);
    return mssvt_launch_status();
}

static int dispatch_block_attn(const AttnPack &pack, int ng, int Cg, int head_dim, int row_capacity, bool kv16, hipStream_t st) {
#define MSSVT_ATTN_CASE(cg, hd)                                   \
    if (Cg == cg && head_dim == hd)                               \
        return launch_block_attn<cg, hd, ((cg / hd + 3) / 4) * 4>(pack, ng, row_capacity, kv16, st);
    MSSVT_ATTN_CASE(8, 8)
    MSSVT_ATTN_CASE(16, 8)
    MSSVT_ATTN_CASE(16, 16)
    MSSVT_ATTN_CASE(24, 8)
    MSSVT_ATTN_CASE(32, 8)
    MSSVT_ATTN_CASE(32, 16)
    MSSVT_ATTN_CASE(32, 32)
    MSSVT_ATTN_CASE(48, 16)
    MSSVT_ATTN_CASE(64, 8)
    MSSVT_ATTN_CASE(64, 16)
    MSSVT_ATTN_CASE(64, 32)
    return MSSVT_E_TOOLARGE;  // shape not instantiated: the caller falls back to the operator path
#undef MSSVT_ATTN_CASE
}

static int block_attention_impl(
    int C, int num_groups, const int *host_c0, const int *host_cg, const int *host_heads, int head_dim, float scale,
    int nq, int key_num_sample, const float *xhat, const int *num_active_dev, const int *perm, const int *q_off,
    const int *nq_valid, const int *num_rows_dev, int row_capacity, const float *qrow_meta, const int *qrow_src,
    const float *const *host_kmeta, const float *wcentre, const float *const *host_Wq, const float *const *host_bq,
    const float *const *host_Wkv, const float *const *host_bkv, const float *const *host_Wo,
    const float *const *host_bo, const float *Wpos, const float *bpos, float *qbuf, float *attn, bool kv16,
    const void *const *host_packed, void *stream) {
    if (!host_c0 || !host_cg || !host_heads || !xhat || !num_active_dev || !perm || !q_off || !nq_valid ||
        !num_rows_dev || !qrow_meta || !qrow_src || !host_kmeta || !wcentre || !host_Wq || !host_bq || !host_Wkv ||
        !host_bkv || !host_Wo || !host_bo || !Wpos || !bpos || !qbuf || !attn || C <= 0 || num_groups <= 0 ||
        head_dim <= 0 || nq <= 0 || key_num_sample <= 0 || row_capacity <= 0)
        return MSSVT_E_BADARG;
    // 16-byte row segments: channel offsets must be float4 aligned
    if (C & 3) return MSSVT_E_BADARG;
    if (key_num_sample > MSSVT_WAVE || (head_dim & 3)) return MSSVT_E_TOOLARGE;
    hipStream_t st = (hipStream_t)stream;
    bool same = num_groups <= ATTN_MAX_GROUPS;
    for (int g = 0; g < num_groups; ++g) same = same && host_cg[g] == host_cg[0];
    AttnPack pack;
    size_t qoff = 0;  // each group's region of qbuf: row_capacity x 4*ceil(heads/4)*Cg floats
    for (int g = 0; g < num_groups; ++g) {
        const int c0 = host_c0[g], Cg = host_cg[g], heads = host_heads[g];
        if (!host_kmeta[g] || !host_Wq[g] || !host_bq[g] || !host_Wkv[g] || !host_bkv[g] || !host_Wo[g] || !host_bo[g])
            return MSSVT_E_BADARG;
        if (Cg <= 0 || heads <= 0 || Cg != heads * head_dim || c0 < 0 || c0 + Cg > C || (c0 & 3)) return MSSVT_E_BADARG;
        // one channel per lane, <= 8 heads per group
        if (Cg > MSSVT_WAVE || heads > 8) return MSSVT_E_TOOLARGE;
        AttnArgs a;
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
        a.qbuf = qbuf + qoff;
        a.attn = attn;
        a.row_capacity = row_capacity;
        a.packed = host_packed ? host_packed[g] : nullptr;
        a.xcd = mssvt_xcd_remap();
        qoff += (size_t)row_capacity * (((heads + 3) / 4) * 4) * Cg;
        if (same) {
            pack.g[g] = a;
        } else {  // unequal group widths: one launch triple per group
            AttnPack one;
            for (int i = 0; i < ATTN_MAX_GROUPS; ++i) one.g[i] = a;
            const int rc = dispatch_block_attn(one, 1, Cg, head_dim, row_capacity, kv16, st);
            if (rc) return rc;
        }
    }
    if (!same) return MSSVT_OK;
    for (int g = num_groups; g < ATTN_MAX_GROUPS; ++g) pack.g[g] = pack.g[0];
    return dispatch_block_attn(pack, num_groups, host_cg[0], head_dim, row_capacity, kv16, st);
}

#define ATTN_ENTRY_PARAMS                                                                                              \
    int C, int num_groups, const int *host_c0, const int *host_cg, const int *host_heads, int head_dim, float scale,   \
    int nq, int key_num_sample, const float *xhat, const int *num_active_dev, const int *perm, const int *q_off,       \
    const int *nq_valid, const int *num_rows_dev, int row_capacity, const float *qrow_meta, const int *qrow_src,       \
    const float *const *host_kmeta, const float *wcentre, const float *const *host_Wq, const float *const *host_bq,    \
    const float *const *host_Wkv, const float *const *host_bkv, const float *const *host_Wo,                           \
    const float *const *host_bo, const float *Wpos, const float *bpos, float *qbuf, float *attn, void *stream
#define ATTN_ENTRY_ARGS                                                                                                \
    C, num_groups, host_c0, host_cg, host_heads, head_dim, scale, nq, key_num_sample, xhat, num_active_dev, perm,      \
    q_off, nq_valid, num_rows_dev, row_capacity, qrow_meta, qrow_src, host_kmeta, wcentre, host_Wq, host_bq, host_Wkv, \
    host_bkv, host_Wo, host_bo, Wpos, bpos, qbuf, attn
extern "C" int mssvt_block_attention(ATTN_ENTRY_PARAMS) { return block_attention_impl(ATTN_ENTRY_ARGS, false, nullptr, stream); }
// launch B with split-fp16 matrix operands (k_attn_kvh) where the shape allows (Cg % 32 == 0, 16 < K <= 64), the fp32
// form otherwise; with host_packed (one mssvt_attn_pack_weights blob per group; head_dim 16) launches A and C run on the
// pre-split fragments too.  The CALLER guarantees the fp16 range of key tokens, Q', Qt, Xbar and V
extern "C" int mssvt_block_attention_kv16(
    int C, int num_groups, const int *host_c0, const int *host_cg, const int *host_heads, int head_dim, float scale,
    int nq, int key_num_sample, const float *xhat, const int *num_active_dev, const int *perm, const int *q_off,
    const int *nq_valid, const int *num_rows_dev, int row_capacity, const float *qrow_meta, const int *qrow_src,
    const float *const *host_kmeta, const float *wcentre, const float *const *host_Wq, const float *const *host_bq,
    const float *const *host_Wkv, const float *const *host_bkv, const float *const *host_Wo,
    const float *const *host_bo, const float *Wpos, const float *bpos, float *qbuf, float *attn,
    const void *const *host_packed, void *stream) {
    return block_attention_impl(ATTN_ENTRY_ARGS, true, host_packed, stream);
}

extern "C" long long mssvt_attn_packed_bytes(int Cg, int head_dim) {
    if (head_dim != 16) return 0;
    if (Cg == 64) return AttnBlob<64>::BYTES;
    if (Cg == 32) return AttnBlob<32>::BYTES;
    return 0;
}

extern "C" int mssvt_attn_pack_weights(int Cg, int head_dim, float scale, const float *Wq, const float *Wkv, const float *Wo,
                                       void *packed, void *stream) {
    if (!Wq || !Wkv || !Wo || !packed) return MSSVT_E_BADARG;
    if (head_dim != 16) return MSSVT_E_TOOLARGE;
    hipStream_t st = (hipStream_t)stream;
    if (Cg == 64) k_attn_pack<64><<<dim3(16, 6), MSSVT_WAVE, 0, st>>>(Wq, Wkv, Wo, scale, reinterpret_cast<char *>(packed));
    else if (Cg == 32) k_attn_pack<32><<<dim3(4, 6), MSSVT_WAVE, 0, st>>>(Wq, Wkv, Wo, scale, reinterpret_cast<char *>(packed));
    else return MSSVT_E_TOOLARGE;
    return mssvt_launch_status();
}

// cell centre in metres, one rounding per op like the reference's torch expression
// (ref: with_coords, mssvt_backbone.py:132-137)
__device__ __forceinline__ float centre_of(int idx, float cell, float lo) {
    return __fadd_rn(__fmul_rn(__fadd_rn((float)idx, 0.5f), cell), lo);
}

// ---------------------------------------------------------------------------------
// interpolation (3-NN, inverse distance) + scatter + first residual
// ---------------------------------------------------------------------------------
struct ScatterArgs {
    int C, nq, n1, interp;
    const float *attn, *x_in;
    float *x_new;
    const int *indices, *win_ind, *num_wins, *win_vstart, *q_ind, *upd_ind, *owner;
    float vsx, vsy, vsz, minx, miny, minz;
    // table mode (tab_row != null): nothing is gathered; per owned voxel the three attention
    // rows and weights are recorded so that a consumer (the fused FFN) can apply them
    int4 *tab_row;
    float4 *tab_w;
    int zero_row;  // row of `attn` that holds zeros: target of empty slots / zero weights
};

#define SC_WPB 4
#define SC_MAXQ 256
#define SC_MAX_SETS 4
struct ScatterPack {
    ScatterArgs s[SC_MAX_SETS];
};

// blockIdx.y = set: the interpolation tables of all (cbs_pattern, interp) variants of a plan in one launch
__global__ void __launch_bounds__(SC_WPB *MSSVT_WAVE) k_block_scatter(ScatterPack pack) {
    const ScatterArgs &a = pack.s[blockIdx.y];
    __shared__ float kx[SC_WPB][SC_MAXQ], ky[SC_WPB][SC_MAXQ], kz[SC_WPB][SC_MAXQ];
    __shared__ int kvalid[SC_WPB][SC_MAXQ];
    const int wv = threadIdx.x / MSSVT_WAVE, lane = lane_id();
    const int nw = *a.num_wins;
    for (int w = blockIdx.x * SC_WPB + wv; w < nw; w += gridDim.x * SC_WPB) {
        const int vstart = a.win_vstart[w];
        if (!a.interp) {  // ref mssvt_backbone.py:327-330: only the query voxels are updated
            for (int i = 0; i < a.nq; ++i) {
                const int v = a.q_ind[(size_t)w * a.nq + i];
                if (v < 0 || a.owner[vstart + v] != w * a.nq + i) continue;
                if (a.tab_row) {
                    if (lane == 0) {
                        a.tab_row[vstart + v] = make_int4(w * a.nq + i, a.zero_row, a.zero_row, 0);
                        a.tab_w[vstart + v] = make_float4(1.f, 0.f, 0.f, 0.f);
                    }
                    continue;
                }
                const float *src = a.attn + ((size_t)w * a.nq + i) * a.C;
                const size_t row = (size_t)(vstart + v) * a.C;
                for (int c = lane; c < a.C; c += MSSVT_WAVE) a.x_new[row + c] = src[c] + a.x_in[row + c];
            }
            continue;
        }
        // known points = ALL nq query slots; empty slots sit at the world origin with zero
        // features (ref :302 gathers coordinates with -1 -> 0 fill) -- kept as is
        // Candidate list for the 3-NN search, in slot order: every valid slot, and of the EMPTY slots only the
        // first three -- all empty slots are the same point (the origin), the search keeps the first seen on
        // ties (strict <), so a fourth one can never enter the best three.  ~5 candidates instead of nq.
        int ncand = 0, nempty = 0;
        for (int i0 = 0; i0 < a.nq; i0 += MSSVT_WAVE) {
            const int i = i0 + lane;
            int v = -1;
            float x = 0.f, y = 0.f, z = 0.f;
            if (i < a.nq) {
                v = a.q_ind[(size_t)w * a.nq + i];
                if (v >= 0) {
                    const int4 vi = reinterpret_cast<const int4 *>(a.indices)[vstart + v];
                    x = centre_of(vi.w, a.vsx, a.minx);
                    y = centre_of(vi.z, a.vsy, a.miny);
                    z = centre_of(vi.y, a.vsz, a.minz);
                }
            }
            const bool empty = i < a.nq && v < 0;
            const unsigned long long me = __ballot(empty);
            const bool keep = i < a.nq && (v >= 0 || nempty + __popcll(me & ((1ull << lane) - 1ull)) < 3);
            const unsigned long long mk = __ballot(keep);
            if (keep) {
                const int p = ncand + __popcll(mk & ((1ull << lane) - 1ull));
                kx[wv][p] = x; ky[wv][p] = y; kz[wv][p] = z;
                kvalid[wv][p] = (i << 1) | (v >= 0 ? 1 : 0);  // original slot, valid bit
            }
            ncand += __popcll(mk);
            nempty += __popcll(me);
        }
        wave_lds_sync();
        for (int s0 = 0; s0 < a.n1; s0 += MSSVT_WAVE) {
            const int s = s0 + lane;
            int v = -1;
            if (s < a.n1) {
                v = a.upd_ind[(size_t)w * a.n1 + s];
                if (v >= 0 && a.owner[vstart + v] != w * a.n1 + s) v = -1;  // another slot owns this voxel
            }
            int i1 = 0, i2 = 0, i3 = 0;
            float w1 = 0.f, w2 = 0.f, w3 = 0.f;
            if (v >= 0) {  // K9 (ref interpolate_gpu.cu:16-59) + weights (ref mssvt_backbone.py:305-307)
                const int4 vi = reinterpret_cast<const int4 *>(a.indices)[vstart + v];
                const float ux = centre_of(vi.w, a.vsx, a.minx), uy = centre_of(vi.z, a.vsy, a.miny),
                            uz = centre_of(vi.y, a.vsz, a.minz);
                // the reference keeps the running bests in double (initial 1e40) and compares float distances
                // against them: the same order as float compares against +inf
                float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
                int c1 = -1, c2 = -1, c3 = -1;  // candidate positions
                for (int k = 0; k < ncand; ++k) {
                    const float dx = ux - kx[wv][k], dy = uy - ky[wv][k], dz = uz - kz[wv][k];
                    const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    if (d < b1) { b3 = b2; c3 = c2; b2 = b1; c2 = c1; b1 = d; c1 = k; }
                    else if (d < b2) { b3 = b2; c3 = c2; b2 = d; c2 = k; }
                    else if (d < b3) { b3 = d; c3 = k; }
                }
                // fewer than three candidates (nq < 3): the reference leaves index 0 / distance 1e40 -> weight ~0
                const int m1 = c1 >= 0 ? kvalid[wv][c1] : 0, m2 = c2 >= 0 ? kvalid[wv][c2] : 0,
                          m3 = c3 >= 0 ? kvalid[wv][c3] : 0;
                i1 = m1 >> 1; i2 = m2 >> 1; i3 = m3 >> 1;
                const float d1 = fmaxf(c1 >= 0 ? sqrtf(b1) : INFINITY, 1e-10f), d2 = fmaxf(c2 >= 0 ? sqrtf(b2) : INFINITY, 1e-10f),
                            d3 = fmaxf(c3 >= 0 ? sqrtf(b3) : INFINITY, 1e-10f);
                w1 = 1.0f / d1; w2 = 1.0f / d2; w3 = 1.0f / d3;
                const float norm = (w1 + w2) + w3;
                w1 /= norm; w2 /= norm; w3 /= norm;
                if (!(m1 & 1) || c1 < 0) w1 = 0.f;  // empty slots carry zero features
                if (!(m2 & 1) || c2 < 0) w2 = 0.f;
                if (!(m3 & 1) || c3 < 0) w3 = 0.f;
            }
            if (a.tab_row) {
                if (v >= 0) {
                    a.tab_row[vstart + v] = make_int4(w1 != 0.f ? w * a.nq + i1 : a.zero_row,
                                                      w2 != 0.f ? w * a.nq + i2 : a.zero_row,
                                                      w3 != 0.f ? w * a.nq + i3 : a.zero_row, 0);
                    a.tab_w[vstart + v] = make_float4(w1, w2, w3, 0.f);
                }
                continue;
            }
            unsigned long long todo = __ballot(v >= 0);
            while (todo) {  // one covered voxel at a time, lanes sweep its channels
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int vv = __shfl(v, src);
                const int j1 = __shfl(i1, src), j2 = __shfl(i2, src), j3 = __shfl(i3, src);
                const float f1 = __shfl(w1, src), f2 = __shfl(w2, src), f3 = __shfl(w3, src);
                const float *r1 = a.attn + ((size_t)w * a.nq + j1) * a.C;
                const float *r2 = a.attn + ((size_t)w * a.nq + j2) * a.C;
                const float *r3 = a.attn + ((size_t)w * a.nq + j3) * a.C;
                const size_t row = (size_t)(vstart + vv) * a.C;
                for (int c = lane; c < a.C; c += MSSVT_WAVE) {
                    float acc = 0.f;  // a zero weight never touches the (unwritten) row of an empty slot
                    if (f1 != 0.f) acc = r1[c] * f1;
                    if (f2 != 0.f) acc += r2[c] * f2;
                    if (f3 != 0.f) acc += r3[c] * f3;
                    a.x_new[row + c] = acc + a.x_in[row + c];
                }
            }
        }
        wave_lds_sync();
    }
}

extern "C" int mssvt_block_interp_scatter(int C, int nq, int n_upd, int use_interpolation,
                                          const float *attn, const float *x_in, float *x_new,
                                          const int *indices, const int *win_ind,
                                          const int *num_wins_dev, int win_capacity,
                                          const int *win_vstart, const int *q_ind,
                                          const int *upd_ind, const int *owner,
                                          const float *host_voxel_size3,
                                          const float *host_range_min3, void *stream) {
    if (!attn || !x_in || !x_new || !indices || !win_ind || !num_wins_dev || !win_vstart || !q_ind ||
        !owner || !host_voxel_size3 || !host_range_min3 || C <= 0 || nq <= 0)
        return MSSVT_E_BADARG;
    if (use_interpolation && (!upd_ind || n_upd <= 0)) return MSSVT_E_BADARG;
    if (nq > SC_MAXQ) return MSSVT_E_TOOLARGE;
    if (win_capacity <= 0) return MSSVT_OK;
    ScatterArgs a;
    a.C = C; a.nq = nq; a.n1 = n_upd; a.interp = use_interpolation;
    a.attn = attn; a.x_in = x_in; a.x_new = x_new;
    a.indices = indices; a.win_ind = win_ind; a.num_wins = num_wins_dev; a.win_vstart = win_vstart;
    a.q_ind = q_ind; a.upd_ind = upd_ind; a.owner = owner;
    a.vsx = host_voxel_size3[0]; a.vsy = host_voxel_size3[1]; a.vsz = host_voxel_size3[2];
    a.minx = host_range_min3[0]; a.miny = host_range_min3[1]; a.minz = host_range_min3[2];
    a.tab_row = nullptr; a.tab_w = nullptr; a.zero_row = 0;
    int grid = divup(win_capacity, SC_WPB);
    if (grid > 4096) grid = 4096;  // grid-stride over the windows actually present
    ScatterPack pack;
    for (int i = 0; i < SC_MAX_SETS; ++i) pack.s[i] = a;
    k_block_scatter<<<grid, SC_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(pack);
    return mssvt_launch_status();
}

// Table form of mssvt_block_interp_scatter: records, per voxel owned by a list slot, the
// (up to) three attention rows and inverse-distance weights instead of applying them.
extern "C" int mssvt_block_interp_table_multi(int num_sets, const int *host_nq, const int *host_n_upd,
                                              const int *host_interp, const int *indices, const int *win_ind,
                                              const int *num_wins_dev, int win_capacity, const int *win_vstart,
                                              const int *const *host_q_ind, const int *const *host_upd_ind,
                                              const int *const *host_owner, const float *host_voxel_size3,
                                              const float *host_range_min3, const int *host_zero_row,
                                              int *const *host_tab_row, float *const *host_tab_w, void *stream) {
    if (num_sets <= 0 || num_sets > SC_MAX_SETS) return num_sets <= 0 ? MSSVT_E_BADARG : MSSVT_E_TOOLARGE;
    if (!host_nq || !host_n_upd || !host_interp || !indices || !win_ind || !num_wins_dev || !win_vstart ||
        !host_q_ind || !host_upd_ind || !host_owner || !host_voxel_size3 || !host_range_min3 || !host_zero_row ||
        !host_tab_row || !host_tab_w)
        return MSSVT_E_BADARG;
    if (win_capacity <= 0) return MSSVT_OK;
    ScatterPack pack;
    for (int i = 0; i < SC_MAX_SETS; ++i) {
        const int k = i < num_sets ? i : 0;
        if (!host_q_ind[k] || !host_owner[k] || !host_tab_row[k] || !host_tab_w[k] || host_nq[k] <= 0)
            return MSSVT_E_BADARG;
        if (host_interp[k] && (!host_upd_ind[k] || host_n_upd[k] <= 0)) return MSSVT_E_BADARG;
        if (host_nq[k] > SC_MAXQ) return MSSVT_E_TOOLARGE;
        ScatterArgs &a = pack.s[i];
        a.C = 0; a.nq = host_nq[k]; a.n1 = host_n_upd[k]; a.interp = host_interp[k];
        a.attn = nullptr; a.x_in = nullptr; a.x_new = nullptr;
        a.indices = indices; a.win_ind = win_ind; a.num_wins = num_wins_dev; a.win_vstart = win_vstart;
        a.q_ind = host_q_ind[k]; a.upd_ind = host_upd_ind[k]; a.owner = host_owner[k];
        a.vsx = host_voxel_size3[0]; a.vsy = host_voxel_size3[1]; a.vsz = host_voxel_size3[2];
        a.minx = host_range_min3[0]; a.miny = host_range_min3[1]; a.minz = host_range_min3[2];
        a.tab_row = reinterpret_cast<int4 *>(host_tab_row[k]);
        a.tab_w = reinterpret_cast<float4 *>(host_tab_w[k]);
        a.zero_row = host_zero_row[k];
    }
    int grid = divup(win_capacity, SC_WPB);
    if (grid > 16384 / num_sets) grid = 16384 / num_sets;  // ~1 window per wave: the per-window chain is 3 dependent round trips
    k_block_scatter<<<dim3(grid, num_sets), SC_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(pack);
    return mssvt_launch_status();
}

extern "C" int mssvt_block_interp_table(int nq, int n_upd, int use_interpolation, const int *indices,
                                        const int *win_ind, const int *num_wins_dev, int win_capacity,
                                        const int *win_vstart, const int *q_ind, const int *upd_ind,
                                        const int *owner, const float *host_voxel_size3,
                                        const float *host_range_min3, int zero_row, int *tab_row,
                                        float *tab_w, void *stream) {
    return mssvt_block_interp_table_multi(1, &nq, &n_upd, &use_interpolation, indices, win_ind, num_wins_dev,
                                          win_capacity, win_vstart, &q_ind, &upd_ind, &owner, host_voxel_size3,
                                          host_range_min3, &zero_row, &tab_row, &tab_w, stream);
}
