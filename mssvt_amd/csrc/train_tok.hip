// train_tok.hip -- the tokens of a Block's attention on compact rows, forward and backward (training path, SURVEY.md
// section 8 f3).
//
//     tok[r][c] = x̂[rows[r]][c0 + c]  +  relu( b[c0 + c] + sum_{j < 6} W[c0 + c][j] * geo[r][j] )        c < cg
//
// = the gathered voxel feature plus the positional embedding of (offset to the window centre, window centre)
// (ref mssvt_backbone.py:43-47 pos_proj = Conv1d(6, C, 1) + ReLU, :270-285 the token sums; the padded gathers K5 /
// K10: group_features_gpu.cu:52-84, group_points_gpu.cu:56-91).  Autograd builds this from a column slice, a gather, a
// concatenation, a zero pad of the 6-wide weight, a library GEMM with K = 8, a clamp and an add -- nine launches
// forward, more backward, per token set (a Block has four: queries and keys of two head groups).  Here: one launch
// forward; backward = the deterministic segmented sum of segment_reduce.hip into a column range of dx̂ (strided entry
// point there) + the weight / bias gradient below.  The ReLU mask is recomputed from `geo` (six FMAs) instead of being
// stored: the same instructions in the same order as the forward, so the sign is the same bit for bit.
//
// Weight gradient: dW[c][j] = sum_r m[r][c] dtok[r][c] geo[r][j], db[c] = sum_r m[r][c] dtok[r][c]; a lane owns four
// channels and walks its rows in ascending order (7 x 4 accumulators), the lane groups of a workgroup are added in group
// order through LDS, the workgroups' slabs in slice order by k_tok_bwd_reduce (which also adds the token sets that share
// channels: the queries cover every channel, each key set its head group's): no atomics, bit-identical run to run.
#include "common.hip.h"

#define TOK_FMA6(P_, W_, GA_, GB_)                                                            \
    P_ = __builtin_fmaf((W_)[0], (GA_).x, P_); P_ = __builtin_fmaf((W_)[1], (GA_).y, P_);     \
    P_ = __builtin_fmaf((W_)[2], (GA_).z, P_); P_ = __builtin_fmaf((W_)[3], (GA_).w, P_);     \
    P_ = __builtin_fmaf((W_)[4], (GB_).x, P_); P_ = __builtin_fmaf((W_)[5], (GB_).y, P_)

template <int LPR>  // lanes per row = cg / 4
__global__ void __launch_bounds__(256) k_tok_fwd(int M, int Csrc, int c0, const int *rows, const float *src, const float4 *geo,
                                                 const float *W6, const float *b, float *tok) {
    constexpr int RPW = MSSVT_WAVE / LPR, CG = 4 * LPR;
    const int lane = lane_id(), sub = lane / LPR, l = lane % LPR;
    const int c = c0 + 4 * l;
    float w[4][6], bb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bb[i] = b[c + i];
#pragma unroll
        for (int j = 0; j < 6; ++j) w[i][j] = W6[(c + i) * 6 + j];
    }
    const int nw = gridDim.x * 4, wave = blockIdx.x * 4 + threadIdx.x / MSSVT_WAVE;
    for (int r = wave * RPW + sub; r < M; r += nw * RPW) {
        const float4 g0 = geo[2 * r], g1 = geo[2 * r + 1];
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src) x = *reinterpret_cast<const float4 *>(src + (size_t)rows[r] * Csrc + c);
        float p0 = bb[0], p1 = bb[1], p2 = bb[2], p3 = bb[3];
        TOK_FMA6(p0, w[0], g0, g1); TOK_FMA6(p1, w[1], g0, g1); TOK_FMA6(p2, w[2], g0, g1); TOK_FMA6(p3, w[3], g0, g1);
        x.x += fmaxf(p0, 0.f); x.y += fmaxf(p1, 0.f); x.z += fmaxf(p2, 0.f); x.w += fmaxf(p3, 0.f);
        *reinterpret_cast<float4 *>(tok + (size_t)r * CG + 4 * l) = x;
    }
}

// slab layout: [slice][cg][8] floats (j < 6: dW, 6: db, 7: unused)
template <int LPR>
__global__ void __launch_bounds__(256) k_tok_bwd_partial(int M, int c0, const float4 *geo, const float *W6, const float *b,
                                                         const float *dtok, int rows_per_slice, float *slab) {
    constexpr int CG = 4 * LPR, GROUPS = 256 / LPR;
    __shared__ float red[28 * 256];
    const int l = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const int c = c0 + 4 * l;
    float w[4][6], bb[4], acc[4][7];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bb[i] = b[c + i];
#pragma unroll
        for (int j = 0; j < 6; ++j) w[i][j] = W6[(c + i) * 6 + j];
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = 0.f;
    }
    const int m0 = blockIdx.x * rows_per_slice, m1 = min(M, m0 + rows_per_slice);
    for (int r = m0 + grp; r < m1; r += GROUPS) {
        const float4 g0 = geo[2 * r], g1 = geo[2 * r + 1];
        const float4 d = *reinterpret_cast<const float4 *>(dtok + (size_t)r * CG + 4 * l);
        const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float p = bb[i];
            TOK_FMA6(p, w[i], g0, g1);
            const float v = p > 0.f ? dd[i] : 0.f;
            acc[i][0] = __builtin_fmaf(v, g0.x, acc[i][0]); acc[i][1] = __builtin_fmaf(v, g0.y, acc[i][1]);
            acc[i][2] = __builtin_fmaf(v, g0.z, acc[i][2]); acc[i][3] = __builtin_fmaf(v, g0.w, acc[i][3]);
            acc[i][4] = __builtin_fmaf(v, g1.x, acc[i][4]); acc[i][5] = __builtin_fmaf(v, g1.y, acc[i][5]);
            acc[i][6] += v;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) red[(i * 7 + j) * 256 + threadIdx.x] = acc[i][j];
    __syncthreads();
    float *out = slab + (size_t)blockIdx.x * CG * 8;
    for (int v = threadIdx.x; v < LPR * 28; v += 256) {
        const int ll = v % LPR, k = v / LPR;  // k = i * 7 + j
        float s = 0.f;
        for (int gq = 0; gq < GROUPS; ++gq) s += red[k * 256 + gq * LPR + ll];
        out[(4 * ll + k / 7) * 8 + k % 7] = s;
    }
}

struct TokSlabs {
    const float *slab[4];
    int slices[4], c0[4], cg[4];
    int n;
};

// one wave per channel: lane = 8 * (slice group) + j; slices of a group in ascending order, the groups in group order,
// the token sets in argument order
__global__ void __launch_bounds__(256) k_tok_bwd_reduce(int Ctot, TokSlabs a, float *dW, float *db) {
    const int lane = lane_id(), j = lane & 7, sg = lane >> 3;
    const int c = blockIdx.x * 4 + threadIdx.x / MSSVT_WAVE;
    if (c >= Ctot) return;
    float total = 0.f;
    for (int k = 0; k < a.n; ++k) {
        if (c < a.c0[k] || c >= a.c0[k] + a.cg[k]) continue;  // wave-uniform
        const float *p = a.slab[k] + (size_t)(c - a.c0[k]) * 8 + j;
        float v = 0.f;
#pragma unroll 4
        for (int s = sg; s < a.slices[k]; s += 8) v += p[(size_t)s * a.cg[k] * 8];
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += __shfl(v, 8 * q + j);
        total += sum;
    }
    if (sg == 0) {
        if (j < 6) dW[(size_t)c * 6 + j] = total;
        else if (j == 6 && db) db[c] = total;
    }
}

static int tok_slices(int M, int cg) {
    const int groups = 256 / (cg / 4);
    int s = divup(M, (long long)groups * 8);  // >= 8 rows per lane group
    return s < 1 ? 1 : (s > 512 ? 512 : s);
}

extern "C" int mssvt_train_tok_forward(int M, int C, int c0, int cg, const int *rows, const float *src, const float *geo8,
                                       const float *W6, const float *b, float *tok, void *stream) {
    if (M < 0 || c0 < 0 || cg <= 0 || (c0 & 3) || !geo8 || !W6 || !b || !tok || (src && (!rows || C < c0 + cg || (C & 3))))
        return MSSVT_E_BADARG;
    if (M == 0) return MSSVT_OK;
    hipStream_t st = (hipStream_t)stream;
#define TOK_FWD(lpr)                                                                                                   \
    case 4 * lpr: {                                                                                                    \
        const int rpw = MSSVT_WAVE / lpr;                                                                               \
        k_tok_fwd<lpr><<<divup(M, rpw * 16), 256, 0, st>>>(M, C, c0, rows, src, (const float4 *)geo8, W6, b, tok);      \
        break;                                                                                                         \
    }
    switch (cg) {
        TOK_FWD(4) TOK_FWD(8) TOK_FWD(16) TOK_FWD(32) TOK_FWD(64)
        default: return MSSVT_E_BADARG;
    }
#undef TOK_FWD
    return mssvt_launch_status();
}

extern "C" long long mssvt_train_tok_slab_floats(int M, int cg) {
    if (M <= 0 || cg <= 0 || cg > 256) return 0;
    return (long long)tok_slices(M, cg) * cg * 8;
}

extern "C" int mssvt_train_tok_backward_partial(int M, int c0, int cg, const float *geo8, const float *W6, const float *b,
                                                const float *dtok, float *slab, void *stream) {
    if (M <= 0 || c0 < 0 || cg <= 0 || (c0 & 3) || !geo8 || !W6 || !b || !dtok || !slab) return MSSVT_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int slices = tok_slices(M, cg);
    const int rps = divup(M, slices);
#define TOK_BWD(lpr)                                                                                                    \
    case 4 * lpr:                                                                                                       \
        k_tok_bwd_partial<lpr><<<slices, 256, 0, st>>>(M, c0, (const float4 *)geo8, W6, b, dtok, rps, slab);             \
        break;
    switch (cg) {
        TOK_BWD(4) TOK_BWD(8) TOK_BWD(16) TOK_BWD(32) TOK_BWD(64)
        default: return MSSVT_E_BADARG;
    }
#undef TOK_BWD
    return mssvt_launch_status();
}

extern "C" int mssvt_train_tok_backward_reduce(int C, int num_sets, const int *host_M, const int *host_c0, const int *host_cg,
                                               const float *const *host_slabs, float *dW, float *db, void *stream) {
    if (C <= 0 || num_sets < 0 || num_sets > 4 || !dW || (num_sets && (!host_M || !host_c0 || !host_cg || !host_slabs)))
        return MSSVT_E_BADARG;
    TokSlabs a;
    a.n = 0;
    for (int k = 0; k < num_sets; ++k) {
        if (host_M[k] <= 0) continue;  // an empty token set adds nothing
        if (!host_slabs[k]) return MSSVT_E_BADARG;
        a.slab[a.n] = host_slabs[k];
        a.slices[a.n] = tok_slices(host_M[k], host_cg[k]);
        a.c0[a.n] = host_c0[k];
        a.cg[a.n] = host_cg[k];
        ++a.n;
    }
    k_tok_bwd_reduce<<<divup(C, 4), 256, 0, (hipStream_t)stream>>>(C, a, dW, db);
    return mssvt_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Compact key sets of a plan (index work of the training path): the valid key slots of every window, window-major, as
// flat arrays -- what ~25 framework launches per scale (mask, nonzero, divisions, gathers, concatenations) built before.
//   k_key_counts : cnt[w] = number of slots of window w whose row id (kmeta[w][k].w as int) is >= 0, 0 for w >= *num_wins;
//                  *total += cnt[w] (integer atomics: the sum does not depend on the order)
//   k_key_compact: slot k of window w goes to position off[w] + (valid slots before k): its voxel row, its window and
//                  the 8 geometry inputs of the positional embedding (offset to the window centre, centre, 0, 0)
// One wave per window, lane = slot (K <= 64).
// ---------------------------------------------------------------------------------------------------------------------
// (the total: one atomicAdd per WORKGROUP of a grid of at most 512 -- an add per window on one address serialises at ~88 per
// microsecond chip-wide: 223 / 373 us per launch at 19k / 33k windows, DESIGN 4 "design rules")
__device__ __forceinline__ void add_block_total(int wave_sum, int *total) {
    __shared__ int part[4];
    if (lane_id() == 0) part[threadIdx.x / MSSVT_WAVE] = wave_sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int s = part[0] + part[1] + part[2] + part[3];
        if (s) atomicAdd(total, s);
    }
}

__global__ void __launch_bounds__(256) k_key_counts(int cap, int K, const int *num_wins, const float4 *kmeta, int *cnt, int *total) {
    const int lane = lane_id(), nw = *num_wins;
    int sum = 0;
    for (int w = blockIdx.x * 4 + threadIdx.x / MSSVT_WAVE; w < cap; w += gridDim.x * 4) {
        int c = 0;
        if (w < nw) {
            const bool ok = lane < K && __builtin_bit_cast(int, kmeta[(size_t)w * K + lane].w) >= 0;
            c = __popcll(__ballot(ok));
        }
        if (lane == 0) cnt[w] = c;
        sum += c;
    }
    add_block_total(sum, total);
}

__global__ void __launch_bounds__(256) k_key_compact(int nw, int K, const float4 *kmeta, const float4 *wcentre, const int *off,
                                                     int *k_rows, int *k_win, float4 *k_geo) {
    const int w = blockIdx.x * 4 + threadIdx.x / MSSVT_WAVE, lane = lane_id();
    if (w >= nw) return;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
    bool ok = false;
    if (lane < K) {
        m = kmeta[(size_t)w * K + lane];
        ok = __builtin_bit_cast(int, m.w) >= 0;
    }
    const unsigned long long mask = __ballot(ok);
    if (!ok) return;
    const int pos = off[w] + __popcll(mask & ((1ull << lane) - 1ull));
    const float4 c = wcentre[w];
    k_rows[pos] = __builtin_bit_cast(int, m.w);
    k_win[pos] = w;
    k_geo[2 * (size_t)pos] = make_float4(m.x, m.y, m.z, c.x);
    k_geo[2 * (size_t)pos + 1] = make_float4(c.y, c.z, 0.f, 0.f);
}

extern "C" int mssvt_train_key_counts(int cap, int K, const int *num_wins_dev, const float *kmeta, int *cnt, int *total_dev,
                                      void *stream) {
    if (cap < 0 || K <= 0 || K > 64 || !num_wins_dev || !kmeta || !cnt || !total_dev) return MSSVT_E_BADARG;
    if (cap == 0) return MSSVT_OK;
    k_key_counts<<<min(divup(cap, 4), 512), 256, 0, (hipStream_t)stream>>>(cap, K, num_wins_dev, (const float4 *)kmeta, cnt, total_dev);
    return mssvt_launch_status();
}

extern "C" int mssvt_train_key_compact(int num_wins, int K, const float *kmeta, const float *wcentre, const int *off, int *k_rows,
                                       int *k_win, float *k_geo8, void *stream) {
    if (num_wins < 0 || K <= 0 || K > 64 || !kmeta || !wcentre || !off || !k_rows || !k_win || !k_geo8) return MSSVT_E_BADARG;
    if (num_wins == 0) return MSSVT_OK;
    k_key_compact<<<divup(num_wins, 4), 256, 0, (hipStream_t)stream>>>(num_wins, K, (const float4 *)kmeta, (const float4 *)wcentre, off,
                                                                       k_rows, k_win, (float4 *)k_geo8);
    return mssvt_launch_status();
}

// The interpolation / scatter table of a Block (mssvt_block_interp_table: per voxel three rows of the padded attention
// buffer + weights) in compact form: idx3 / w3 (3 per voxel) name compact attention rows through `inv` (padded row ->
// compact row); a voxel the attention does not update (tab_row[v].x < 0) gets the zero row R with weight 0.
__global__ void __launch_bounds__(256) k_interp_compact(int N, int R, const int *inv, const int4 *tab_row, const float4 *tab_w, int *idx3,
                                                        float *w3, unsigned char *owned) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const int4 t = tab_row[v];
    const float4 w = tab_w[v];
    const bool own = t.x >= 0;
    owned[v] = own ? 1 : 0;
    idx3[3 * v + 0] = own ? inv[max(t.x, 0)] : R; idx3[3 * v + 1] = own ? inv[max(t.y, 0)] : R; idx3[3 * v + 2] = own ? inv[max(t.z, 0)] : R;
    w3[3 * v + 0] = own ? w.x : 0.f; w3[3 * v + 1] = own ? w.y : 0.f; w3[3 * v + 2] = own ? w.z : 0.f;
}

extern "C" int mssvt_train_interp_compact(int N, int R, const int *inv, const int *tab_row4, const float *tab_w4, int *idx3, float *w3,
                                          unsigned char *owned, void *stream) {
    if (N < 0 || R < 0 || !inv || !tab_row4 || !tab_w4 || !idx3 || !w3 || !owned) return MSSVT_E_BADARG;
    if (N == 0) return MSSVT_OK;
    k_interp_compact<<<divup(N, 256), 256, 0, (hipStream_t)stream>>>(N, R, inv, (const int4 *)tab_row4, (const float4 *)tab_w4, idx3, w3,
                                                                     owned);
    return mssvt_launch_status();
}

// Compact (window, voxel) pairs of a CompressBlock's window lists (index work of the training path): k_ind (nw, ns) int32
// = positions inside the window's run of voxel rows, < 0: empty (mssvt_window_plan_one).  _counts as for the key sets;
// _compact writes, at the prefix position, the voxel row win_vstart[w] + k, the window, and the geometry inputs of the
// positional embedding: voxel centre - window centre, window centre ((index + 0.5) * cell + min, three separately rounded
// fp32 operations as in ref with_coords, mssvt_backbone.py:132-137), 0, 0.
__global__ void __launch_bounds__(256) k_list_counts(int nw, int ns, const int *k_ind, int *cnt, int *total) {
    const int lane = lane_id();
    int sum = 0;
    for (int w = blockIdx.x * 4 + threadIdx.x / MSSVT_WAVE; w < nw; w += gridDim.x * 4) {
        const bool ok = lane < ns && k_ind[(size_t)w * ns + lane] >= 0;
        const int c = __popcll(__ballot(ok));
        if (lane == 0) cnt[w] = c;
        sum += c;
    }
    add_block_total(sum, total);
}

struct PairGeo {
    float vs[3], mn[3], ws[3];
};
__global__ void __launch_bounds__(256) k_pairs_compact(int nw, int ns, const int *k_ind, const int *win_vstart, const int *off,
                                                       const int4 *indices, const int4 *win_ind, PairGeo gq, int *pair_vox, int *pair_win,
                                                       float4 *geo) {
    const int w = blockIdx.x * 4 + threadIdx.x / MSSVT_WAVE, lane = lane_id();
    if (w >= nw) return;
    const int k = lane < ns ? k_ind[(size_t)w * ns + lane] : -1;
    const bool ok = k >= 0;
    const unsigned long long mask = __ballot(ok);
    if (!ok) return;
    const int pos = off[w] + __popcll(mask & ((1ull << lane) - 1ull));
    const int v = win_vstart[w] + k;
    const int4 vi = indices[v], wi = win_ind[w];  // (b, z, y, x)
    const float vx = ((float)vi.w + 0.5f) * gq.vs[0] + gq.mn[0], vy = ((float)vi.z + 0.5f) * gq.vs[1] + gq.mn[1],
                vz = ((float)vi.y + 0.5f) * gq.vs[2] + gq.mn[2];
    const float cx = ((float)wi.w + 0.5f) * gq.ws[0] + gq.mn[0], cy = ((float)wi.z + 0.5f) * gq.ws[1] + gq.mn[1],
                cz = ((float)wi.y + 0.5f) * gq.ws[2] + gq.mn[2];
    pair_vox[pos] = v;
    pair_win[pos] = w;
    geo[2 * (size_t)pos] = make_float4(vx - cx, vy - cy, vz - cz, cx);
    geo[2 * (size_t)pos + 1] = make_float4(cy, cz, 0.f, 0.f);
}

extern "C" int mssvt_train_list_counts(int num_wins, int ns, const int *k_ind, int *cnt, int *total_dev, void *stream) {
    if (num_wins < 0 || ns <= 0 || ns > 64 || !k_ind || !cnt || !total_dev) return MSSVT_E_BADARG;
    if (num_wins == 0) return MSSVT_OK;
    k_list_counts<<<min(divup(num_wins, 4), 512), 256, 0, (hipStream_t)stream>>>(num_wins, ns, k_ind, cnt, total_dev);
    return mssvt_launch_status();
}

extern "C" int mssvt_train_pairs_compact(int num_wins, int ns, const int *k_ind, const int *win_vstart, const int *off,
                                         const int *indices, const int *win_ind, const float *host_voxel_size3,
                                         const float *host_range_min3, const float *host_win_size3, int *pair_vox, int *pair_win,
                                         float *geo8, void *stream) {
    if (num_wins < 0 || ns <= 0 || ns > 64 || !k_ind || !win_vstart || !off || !indices || !win_ind || !host_voxel_size3 ||
        !host_range_min3 || !host_win_size3 || !pair_vox || !pair_win || !geo8)
        return MSSVT_E_BADARG;
    if (num_wins == 0) return MSSVT_OK;
    PairGeo gq;
    for (int i = 0; i < 3; ++i) {
        gq.vs[i] = host_voxel_size3[i];
        gq.mn[i] = host_range_min3[i];
        gq.ws[i] = host_win_size3[i];
    }
    k_pairs_compact<<<divup(num_wins, 4), 256, 0, (hipStream_t)stream>>>(num_wins, ns, k_ind, win_vstart, off, (const int4 *)indices,
                                                                         (const int4 *)win_ind, gq, pair_vox, pair_win, (float4 *)geo8);
    return mssvt_launch_status();
}

// Compact query rows of a window plan's query pattern (mssvt_plan_order leaves row_meta (rows,4) f32 = offset to the window
// centre xyz + voxel row as int bits, row_src (rows,2) int32 = window, padded attention row): the voxel rows and the eight
// geometry inputs of the positional embedding for the first R rows.
__global__ void __launch_bounds__(256) k_query_sets(int R, const float4 *row_meta, const int2 *row_src, const float4 *wcentre, int *q_rows,
                                                    float4 *q_geo) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float4 m = row_meta[r];
    const float4 c = wcentre[row_src[r].x];
    q_rows[r] = __builtin_bit_cast(int, m.w);
    q_geo[2 * (size_t)r] = make_float4(m.x, m.y, m.z, c.x);
    q_geo[2 * (size_t)r + 1] = make_float4(c.y, c.z, 0.f, 0.f);
}

extern "C" int mssvt_train_query_sets(int R, const float *row_meta, const int *row_src, const float *wcentre, int *q_rows,
                                      float *q_geo8, void *stream) {
    if (R < 0 || !row_meta || !row_src || !wcentre || !q_rows || !q_geo8) return MSSVT_E_BADARG;
    if (R == 0) return MSSVT_OK;
    k_query_sets<<<divup(R, 256), 256, 0, (hipStream_t)stream>>>(R, (const float4 *)row_meta, (const int2 *)row_src,
                                                                 (const float4 *)wcentre, q_rows, (float4 *)q_geo8);
    return mssvt_launch_status();
}

// Inverse of a gather of DISTINCT rows (dst[i] = src[idx[i]], every source row named at most once) as the ranges of a
// segmented sum over the source rows: source row v sums entry bwd_idx[v] if bwd_end[v] == v + 1, nothing if == v.
__global__ void __launch_bounds__(256) k_inverse_scatter(int nnz, const int *idx, int *inv) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nnz) inv[idx[e]] = e;
}
__global__ void __launch_bounds__(256) k_inverse_ranges(int n_src, const int *inv, int *bwd_idx, int *bwd_end) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_src) return;
    const int e = inv[v];
    bwd_idx[v] = e > 0 ? e : 0;
    bwd_end[v] = v + (e >= 0 ? 1 : 0);
}

extern "C" int mssvt_train_unique_inverse(int nnz, int n_src, const int *idx, int *inv_scratch, int *bwd_idx, int *bwd_end, void *stream) {
    if (nnz < 0 || n_src < 0 || (nnz && !idx) || (n_src && (!inv_scratch || !bwd_idx || !bwd_end))) return MSSVT_E_BADARG;
    if (n_src == 0) return MSSVT_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(inv_scratch, 0xFF, (size_t)n_src * 4, st);  // -1: no entry reads the row
    if (e != hipSuccess) return (int)e;
    if (nnz) k_inverse_scatter<<<divup(nnz, 256), 256, 0, st>>>(nnz, idx, inv_scratch);
    k_inverse_ranges<<<divup(n_src, 256), 256, 0, st>>>(n_src, inv_scratch, bwd_idx, bwd_end);
    return mssvt_launch_status();
}
