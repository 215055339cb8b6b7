// compress.hip -- ragged (per-window) parts of the CompressBlock fast path.
//
// The CompressBlock (ref: mssvt_backbone.py:351-398) pools every non-empty window
// into one output voxel: keys = the window's voxels (K4 list), query = channel-wise
// max over the padded key tensor, attention with nq = 1.  The reference materialises
// padded (nw, C, ns) tensors (534 MB each at 160k points) and runs its two-layer
// positional MLP on every padded slot.  Here the padded tensors never exist: the
// valid (window, slot) PAIRS are enumerated once (one row per pair, ~1 per voxel),
// dense per-row math (positional layer 2, K/V projection, output projection, FFN)
// runs as plain library GEMMs over those rows, and the ragged pieces are the four
// HBM-bound kernels below (one wavefront per window / pair row, lanes over channels
// so every feature row is one coalesced segment):
//   k_window_plan_one   K4 list + pair-row allocation            (index work)
//   k_compress_pos1     rel. coordinates + positional layer 1     (R x C rows out)
//   k_compress_pool     query token = max over the window's keys  (nw x C out)
//   k_compress_attn     nq = 1 attention over the window's K/V rows (online softmax)
#include "common.hip.h"

#define CP_WPB 4

__device__ __forceinline__ float cell_centre(int idx, float cell, float lo) {
    return __fadd_rn(__fmul_rn(__fadd_rn((float)idx, 0.5f), cell), lo);  // ref with_coords :132-137
}

// ---------------------------------------------------------------------------------
// plan: K4 (ref ms_sparse_attention_gpu.cu:383-433) + pair rows.
// Every valid (window, slot) pair gets a row in the pair arrays:
//   disjoint lists (odd window sizes, the normal case): row = the voxel's own feature
//     row, pad row of window w = num_voxels + w -- no allocation, no atomics;
//   overlapping lists (even sizes): rows [pair_base[w], +cnt) (+1 pad) are reserved with
//     one atomicAdd per window on counters[0].
// pair_win[row] = window (-1: row unused), pair_vox[row] = voxel feature row (-1: pad).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(CP_WPB *MSSVT_WAVE)
    k_window_plan_one(int x_max, int y_max, int z_max, int x_ws, int y_ws, int z_ws, int max_win1,
                      int hash_size, int n_win1, const int *q_win1, const int *win_indices,
                      const int *num_wins, const slot_t *table, const int *v_bs_cnt, int with_pad,
                      int disjoint, int num_voxels, int *k_ind, int *win_vstart, int *win_cnt,
                      int *pair_base, int *pair_win, int *pair_vox, int *counters,
                      const unsigned long long *occ, const int *col_vbase, const int *level_status) {
    const int w = blockIdx.x * CP_WPB + threadIdx.x / MSSVT_WAVE;
    if (w >= *num_wins) return;
    const int lane = lane_id();
    // sorted voxel list (mssvt_level_setup_sorted): occupancy bit = hit, column base + popcount below = index
    const bool ranked = occ != nullptr && col_vbase != nullptr && !(level_status[0] & ST_UNSORTED);
    const int4 wi = reinterpret_cast<const int4 *>(win_indices)[w];
    const slot_t *tab = table + (size_t)wi.x * hash_size;
    int vstart = 0;
    for (int k = 0; k < wi.x; ++k) vstart += v_bs_cnt[k];
    const int cx = wi.w * x_ws + x_ws / 2, cy = wi.z * y_ws + y_ws / 2, cz = wi.y * z_ws + z_ws / 2;
    for (int k = lane; k < max_win1; k += MSSVT_WAVE) k_ind[(size_t)w * max_win1 + k] = -1;
    int cnt = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    // overlapping lists: pass 0 counts (so the rows can be reserved with one atomic), pass 1 writes
    for (int pass = disjoint ? 1 : 0; pass < 2; ++pass) {
        int base = 0;
        if (pass == 1 && !disjoint) {
            const int total = (cnt < max_win1 ? cnt : max_win1) + (with_pad ? 1 : 0);
            if (lane == 0) base = atomicAdd(counters, total);
            base = __builtin_amdgcn_readfirstlane(base);
            if (lane == 0) {
                pair_base[w] = base;
                if (with_pad) {
                    pair_win[base + total - 1] = w;
                    pair_vox[base + total - 1] = -1;
                }
            }
            cnt = 0;
        }
        for (int bq = 0; bq < n_win1 && cnt < max_win1; bq += MSSVT_WAVE) {
            const int q = bq + lane;
            int sv = MSSVT_EMPTY;
            if (q < n_win1) {
                const int sx = cx + q_win1[q * 3 + 0], sy = cy + q_win1[q * 3 + 1], sz = cz + q_win1[q * 3 + 2];
                if (!(sx >= x_max || sx < 0 || sy >= y_max || sy < 0 || sz >= z_max || sz < 0)) {
                    if (ranked) {
                        const size_t col = ((size_t)wi.x * x_max + sx) * y_max + sy;
                        const unsigned long long word = occ[col];
                        if ((word >> sz) & 1ull) sv = col_vbase[col] + __popcll(word & ((1ull << sz) - 1ull));
                    } else if (table) {
                        sv = table_find(sx * y_max * z_max + sy * z_max + sz, hash_size, tab);
                    }
                }
            }
            const bool hit = sv != MSSVT_EMPTY;
            const unsigned long long m = __ballot(hit);
            if (pass == 1 && hit) {
                const int p = cnt + __popcll(m & below);
                if (p < max_win1) {
                    k_ind[(size_t)w * max_win1 + p] = sv;
                    const int row = disjoint ? vstart + sv : base + p;
                    pair_win[row] = w;
                    pair_vox[row] = vstart + sv;
                }
            }
            cnt += __popcll(m);
        }
    }
    if (lane == 0) {
        win_vstart[w] = vstart;
        win_cnt[w] = cnt < max_win1 ? cnt : max_win1;
        if (disjoint) {
            pair_base[w] = -1;  // rows are addressed through the voxel index instead
            if (with_pad) {
                pair_win[num_voxels + w] = w;
                pair_vox[num_voxels + w] = -1;
            }
        }
    }
}

// The same plan for PILLAR windows of a sorted level (window = one (x, y) column: x_ws = y_ws = 1, every table offset
// inside the column; the [1,1,32] CompressBlock of mssvt.yaml): one LANE per window.  The list is the column's
// occupancy word walked in table order -- no cross-lane work at all, where the wave-per-window form above spends a
// 64-lane wave and four dependent round trips on <= 32 cells (20 us for 7k windows, all of it latency).
__global__ void __launch_bounds__(256)
    k_window_plan_pillars(int x_max, int y_max, int z_max, int z_ws, int max_win1, int n_win1, const int *q_win1,
                          const int *win_indices, const int *num_wins, const int *v_bs_cnt, int *k_ind, int *win_vstart,
                          int *win_cnt, int *pair_base, int *pair_win, int *pair_vox, const unsigned long long *occ,
                          const int *col_vbase, const int *level_status, int win_capacity) {
    // the table's z offsets (<= 64, the caller checked) in the lanes of a register: no load inside the walk
    const int oz_lane = q_win1[min(lane_id(), n_win1 - 1) * 3 + 2];
    const int w = blockIdx.x * 256 + threadIdx.x;
    const int4 wi = reinterpret_cast<const int4 *>(win_indices)[min(w, win_capacity - 1)];  // [b, wz, wy, wx]
    const bool live = w < *num_wins && !(level_status[0] & ST_UNSORTED);
    int vstart = 0;
    for (int k = 0; k < (live ? wi.x : 0); ++k) vstart += v_bs_cnt[k];
    const int cz = wi.y * z_ws + z_ws / 2;
    const size_t col = live ? ((size_t)wi.x * x_max + wi.w) * y_max + wi.z : 0;
    const unsigned long long word = live ? occ[col] : 0ull;
    const int base = col_vbase[col];
    int *row = k_ind + (size_t)w * max_win1;
    int cnt = 0;
    for (int q = 0; q < n_win1; ++q) {  // (every lane of the wave takes part: readlane)
        const int sz = cz + __builtin_amdgcn_readlane(oz_lane, q);
        if (live && (unsigned int)sz < (unsigned int)z_max && ((word >> sz) & 1ull)) {
            if (cnt < max_win1) {
                const int sv = base + __popcll(word & ((1ull << sz) - 1ull));
                row[cnt] = sv;
                pair_win[vstart + sv] = w;
                pair_vox[vstart + sv] = vstart + sv;
            }
            ++cnt;
        }
    }
    if (!live) return;
    const int nk = cnt < max_win1 ? cnt : max_win1;
    for (int k = nk; k < max_win1; ++k) row[k] = -1;
    win_vstart[w] = vstart;
    win_cnt[w] = nk;
    pair_base[w] = -1;
}

// ---------------------------------------------------------------------------------
// positional layer 1 for every pair row: relu(W1 [rel ; centre] + b1), rel = voxel
// centre - window centre (NOT masked in the CompressBlock, ref :372), pad row: the
// slot's coordinates are the origin.  W1 (C,6), out (R,C).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(CP_WPB *MSSVT_WAVE)
    k_compress_pos1(int C, const int *num_rows, const int *pair_win, const int *pair_vox,
                    const int *indices, const int *win_indices, float vsx, float vsy, float vsz,
                    float minx, float miny, float minz, float wsx, float wsy, float wsz,
                    const float *W1, const float *b1, float *out) {
    const int nrows = *num_rows;
    const int lane = lane_id();
    for (int r = blockIdx.x * CP_WPB + threadIdx.x / MSSVT_WAVE; r < nrows; r += gridDim.x * CP_WPB) {
        const int pw = pair_win[r];
        if (pw < 0) {  // voxel in no list (ragged border / truncated list): keep the row finite
            for (int c = lane; c < C; c += MSSVT_WAVE) out[(size_t)r * C + c] = 0.0f;
            continue;
        }
        const int4 wi = reinterpret_cast<const int4 *>(win_indices)[pw];
        const float cxm = cell_centre(wi.w, wsx, minx), cym = cell_centre(wi.z, wsy, miny),
                    czm = cell_centre(wi.y, wsz, minz);
        const int vox = pair_vox[r];
        float px = 0.f, py = 0.f, pz = 0.f;
        if (vox >= 0) {
            const int4 vi = reinterpret_cast<const int4 *>(indices)[vox];
            px = cell_centre(vi.w, vsx, minx);
            py = cell_centre(vi.z, vsy, miny);
            pz = cell_centre(vi.y, vsz, minz);
        }
        const float rx = px - cxm, ry = py - cym, rz = pz - czm;
        for (int c = lane; c < C; c += MSSVT_WAVE) {
            const float *wr = W1 + (size_t)c * 6;
            const float v = b1[c] + wr[0] * rx + wr[1] * ry + wr[2] * rz + wr[3] * cxm + wr[4] * cym + wr[5] * czm;
            out[(size_t)r * C + c] = fmaxf(v, 0.0f);
        }
    }
}

// k_in[r] = xhat[pair_vox[r]] + pos2[r]  (in place on pos2; pad rows keep pos2 only)
__global__ void __launch_bounds__(CP_WPB *MSSVT_WAVE)
    k_compress_add_features(int C, const int *num_rows, const int *pair_vox, const float *xhat, float *pos2) {
    const int nrows = *num_rows;
    const int lane = lane_id();
    for (int r = blockIdx.x * CP_WPB + threadIdx.x / MSSVT_WAVE; r < nrows; r += gridDim.x * CP_WPB) {
        const int vox = pair_vox[r];
        if (vox < 0) continue;
        for (int c = lane; c < C; c += MSSVT_WAVE) pos2[(size_t)r * C + c] += xhat[(size_t)vox * C + c];
    }
}

// query token: max over the (zero padded) key features (ref :370)
__global__ void __launch_bounds__(CP_WPB *MSSVT_WAVE)
    k_compress_pool(int C, int max_win1, const int *num_wins, const int *k_ind, const int *win_vstart,
                    const int *win_cnt, const float *xhat, float *q_tok) {
    const int nw = *num_wins;
    const int lane = lane_id();
    for (int w = blockIdx.x * CP_WPB + threadIdx.x / MSSVT_WAVE; w < nw; w += gridDim.x * CP_WPB) {
        const int cnt = win_cnt[w], vstart = win_vstart[w];
        for (int c = lane; c < C; c += MSSVT_WAVE) {
            float m = cnt < max_win1 ? 0.0f : -INFINITY;  // empty slots contribute zeros
            for (int s = 0; s < cnt; ++s)
                m = fmaxf(m, xhat[(size_t)(vstart + k_ind[(size_t)w * max_win1 + s]) * C + c]);
            q_tok[(size_t)w * C + c] = m;
        }
    }
}

// ---------------------------------------------------------------------------------
// nq = 1 attention per window and head group (ref mssvt_utils.py:112-150, seq-first
// call of :377-381).  qp (nw, C) = projected queries (all groups concatenated);
// kv (R, 2*Cg) = [K | V] rows of THIS group's projection for every pair row.
// Group g attends to list slots [g*nk, (g+1)*nk).  Masked (= empty) slots carry an
// additive -100 in the reference: they are skipped (weight <= e^-100) unless the whole
// slot range is empty, where the softmax is uniform over identical pad tokens and the
// result is the pad row's V exactly.  lane = channel; a head's hd channels are
// hd consecutive lanes (hd | 64), scores are reduced with xor-shuffles inside them.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(CP_WPB *MSSVT_WAVE)
    k_compress_attn(int C, int c0, int Cg, int hd, float scale, int nk, int g, int with_pad, int max_win1,
                    int num_voxels, const int *num_wins, const int *win_cnt, const int *pair_base,
                    const int *k_ind, const int *win_vstart, const float *qp, const float *kv, float *out) {
    const int nw = *num_wins;
    const int lane = lane_id();
    for (int w = blockIdx.x * CP_WPB + threadIdx.x / MSSVT_WAVE; w < nw; w += gridDim.x * CP_WPB) {
        const int cnt = win_cnt[w], base = pair_base[w], vstart = win_vstart[w];
        const int s_lo = g * nk, s_hi = min((g + 1) * nk, cnt);
        // pair row of slot s / of the pad token (see k_window_plan_one)
        const int pad_row = base >= 0 ? base + cnt : num_voxels + w;
        for (int cb = 0; cb < Cg; cb += MSSVT_WAVE) {  // Cg <= 64: one pass
            const int c = cb + lane;
            const bool act = c < Cg;
            const int cc = act ? c : 0;
            float res;
            if (s_hi <= s_lo) {  // every slot of this group is empty
                res = with_pad ? kv[(size_t)pad_row * 2 * Cg + Cg + cc] : 0.0f;
            } else {
                const float q = qp[(size_t)w * C + c0 + cc] * scale;
                float m = -INFINITY, l = 0.0f, acc = 0.0f;
                for (int s = s_lo; s < s_hi; ++s) {
                    const int prow = base >= 0 ? base + s : vstart + k_ind[(size_t)w * max_win1 + s];
                    const float *row = kv + (size_t)prow * 2 * Cg;
                    float sc = act ? q * row[cc] : 0.0f;
                    for (int off = 1; off < hd; off <<= 1) sc += __shfl_xor(sc, off);
                    const float mn = fmaxf(m, sc);
                    const float corr = expf(m - mn), p = expf(sc - mn);
                    l = l * corr + p;
                    acc = acc * corr + p * row[Cg + cc];
                    m = mn;
                }
                res = acc / l;
            }
            if (act) out[(size_t)w * C + c0 + c] = res;
        }
    }
}

// ---- host entry points -------------------------------------------------------------
static inline int stride_blocks(long long items) {
    long long g = (items + CP_WPB - 1) / CP_WPB;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

extern "C" int mssvt_window_plan_one(int x_max, int y_max, int z_max, int x_ws, int y_ws, int z_ws,
                                     int max_num_win1, int hash_size, int num_win1,
                                     const int *vox_query_win1, const int *win_indices,
                                     const int *num_wins_dev, int win_capacity, const int *xyz_to_vidx,
                                     const int *v_bs_cnt, int with_pad, int disjoint_lists,
                                     int num_voxels, int *k_ind, int *win_vstart, int *win_cnt,
                                     int *pair_base, int *pair_win, int *pair_vox, int *counters,
                                     const unsigned long long *occ_columns, const int *column_vbase,
                                     const int *level_status_dev, void *stream) {
    const bool ranked = occ_columns && column_vbase && level_status_dev && z_max <= 64;
    if (!vox_query_win1 || !win_indices || !num_wins_dev || (!xyz_to_vidx && !ranked) || !v_bs_cnt || !k_ind ||
        !win_vstart || !win_cnt || !pair_base || !pair_win || !pair_vox || !counters || hash_size <= 0 ||
        max_num_win1 <= 0)
        return MSSVT_E_BADARG;
    if (win_capacity <= 0) return MSSVT_OK;
    if (!disjoint_lists) {  // the row counter is only touched when rows have to be reserved (overlapping lists)
        hipError_t e = hipMemsetAsync(counters, 0, sizeof(int), (hipStream_t)stream);
        if (e != hipSuccess) return (int)e;
    }
    if (ranked && disjoint_lists == 2 && !with_pad && x_ws == 1 && y_ws == 1 && num_win1 >= 1 && num_win1 <= MSSVT_WAVE) {
        k_window_plan_pillars<<<divup(win_capacity, 256), 256, 0, (hipStream_t)stream>>>(
            x_max, y_max, z_max, z_ws, max_num_win1, num_win1, vox_query_win1, win_indices, num_wins_dev, v_bs_cnt, k_ind,
            win_vstart, win_cnt, pair_base, pair_win, pair_vox, occ_columns, column_vbase, level_status_dev, win_capacity);
        return mssvt_launch_status();
    }
    k_window_plan_one<<<divup(win_capacity, CP_WPB), CP_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(
        x_max, y_max, z_max, x_ws, y_ws, z_ws, max_num_win1, hash_size, num_win1, vox_query_win1,
        win_indices, num_wins_dev, reinterpret_cast<const slot_t *>(xyz_to_vidx), v_bs_cnt, with_pad,
        disjoint_lists, num_voxels, k_ind, win_vstart, win_cnt, pair_base, pair_win, pair_vox, counters,
        ranked ? occ_columns : nullptr, ranked ? column_vbase : nullptr, ranked ? level_status_dev : nullptr);
    return mssvt_launch_status();
}

extern "C" int mssvt_compress_pos1(int C, const int *num_rows_dev, int row_capacity, const int *pair_win,
                                   const int *pair_vox, const int *indices, const int *win_indices,
                                   const float *host_voxel_size3, const float *host_range_min3,
                                   const float *host_win_size3, const float *W1, const float *b1,
                                   float *out, void *stream) {
    if (!num_rows_dev || !pair_win || !pair_vox || !indices || !win_indices || !host_voxel_size3 ||
        !host_range_min3 || !host_win_size3 || !W1 || !b1 || !out || C <= 0)
        return MSSVT_E_BADARG;
    if (row_capacity <= 0) return MSSVT_OK;
    k_compress_pos1<<<stride_blocks(row_capacity), CP_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(
        C, num_rows_dev, pair_win, pair_vox, indices, win_indices, host_voxel_size3[0], host_voxel_size3[1],
        host_voxel_size3[2], host_range_min3[0], host_range_min3[1], host_range_min3[2], host_win_size3[0],
        host_win_size3[1], host_win_size3[2], W1, b1, out);
    return mssvt_launch_status();
}

extern "C" int mssvt_compress_add_features(int C, const int *num_rows_dev, int row_capacity,
                                           const int *pair_vox, const float *xhat, float *rows,
                                           void *stream) {
    if (!num_rows_dev || !pair_vox || !xhat || !rows || C <= 0) return MSSVT_E_BADARG;
    if (row_capacity <= 0) return MSSVT_OK;
    k_compress_add_features<<<stride_blocks(row_capacity), CP_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(
        C, num_rows_dev, pair_vox, xhat, rows);
    return mssvt_launch_status();
}

extern "C" int mssvt_compress_pool(int C, int max_num_win1, const int *num_wins_dev, int win_capacity,
                                   const int *k_ind, const int *win_vstart, const int *win_cnt,
                                   const float *xhat, float *q_tok, void *stream) {
    if (!num_wins_dev || !k_ind || !win_vstart || !win_cnt || !xhat || !q_tok || C <= 0 || max_num_win1 <= 0)
        return MSSVT_E_BADARG;
    if (win_capacity <= 0) return MSSVT_OK;
    k_compress_pool<<<stride_blocks(win_capacity), CP_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(
        C, max_num_win1, num_wins_dev, k_ind, win_vstart, win_cnt, xhat, q_tok);
    return mssvt_launch_status();
}

extern "C" int mssvt_compress_attention_group(int C, int c0, int Cg, int head_dim, float scale,
                                              int keys_per_group, int group, int with_pad,
                                              int max_num_win1, int num_voxels,
                                              const int *num_wins_dev, int win_capacity,
                                              const int *win_cnt, const int *pair_base, const int *k_ind,
                                              const int *win_vstart, const float *qp, const float *kv,
                                              float *out, void *stream) {
    if (!num_wins_dev || !win_cnt || !pair_base || !k_ind || !win_vstart || !qp || !kv || !out || C <= 0 ||
        Cg <= 0 || head_dim <= 0)
        return MSSVT_E_BADARG;
    if (head_dim > MSSVT_WAVE || (MSSVT_WAVE % head_dim) != 0 || (head_dim & (head_dim - 1)) != 0)
        return MSSVT_E_TOOLARGE;  // a head must be a power-of-two run of lanes
    if (win_capacity <= 0) return MSSVT_OK;
    k_compress_attn<<<stride_blocks(win_capacity), CP_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(
        C, c0, Cg, head_dim, scale, keys_per_group, group, with_pad, max_num_win1, num_voxels, num_wins_dev,
        win_cnt, pair_base, k_ind, win_vstart, qp, kv, out);
    return mssvt_launch_status();
}
