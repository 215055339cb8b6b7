// window_gather.hip -- K3 / K4: chessboard + mixed-scale window voxel gather.
//
// Replaces gather_two_window_voxels_with_hash_kernel and
// gather_one_window_voxels_with_hash_kernel
// (ref: mssvt/src/ms_sparse_attention_gpu.cu:193-350, :383-433).  The reference
// walks up to |win2| (343 for 7x7x7) offsets SERIALLY in one thread per window:
// ~20k threads, each a chain of dependent hash probes.  Here ONE WAVEFRONT owns
// a window and its 64 lanes probe 64 offsets at a time; the ordered, truncated
// append the reference does with per-thread counters is reproduced bit-exactly
// with __ballot + prefix popcount (lane order == table order).
//
// List membership (ref :239-259, :275-295, :311-324, :339-345):
//   odd   <- hits of the odd table
//   even  <- hits of the even table
//   win1  <- hits of odd, even, win1_other (in that order)
//   win2  <- hits of all four tables (in that order)
// each truncated at its max_num_*.  The reference's early `return`s only skip
// work once every still-open list is full, so they do not change the result.
#include "common.hip.h"

#define WAVES_PER_BLOCK 4

struct TwoWinArgs {
    int x_max, y_max, z_max, x_ws, y_ws, z_ws;
    int max_odd, max_even, max_win1, max_win2;
    int num_wins, hash_size;
    int n_odd, n_even, n_win1, n_win2;
    int *ind_odd, *ind_even, *ind_win1, *ind_win2;
    int *c_odd, *c_even, *c_win1, *c_win2;
    const int *q_odd, *q_even, *q_win1, *q_win2;
    const int *win_indices;
    const slot_t *table;
};

__device__ __forceinline__ void put(int *ind, int *coord, size_t row, int max_num, int pos, int sv,
                                    int ox, int oy, int oz) {
    if (pos < max_num) {
        ind[row * max_num + pos] = sv;
        int *c = coord + (row * max_num + pos) * 3;
        c[0] = ox;
        c[1] = oy;
        c[2] = oz;
    }
}

__global__ void __launch_bounds__(WAVES_PER_BLOCK *MSSVT_WAVE) k_gather_two_window(TwoWinArgs a) {
    const int w = blockIdx.x * WAVES_PER_BLOCK + threadIdx.x / MSSVT_WAVE;
    if (w >= a.num_wins) return;  // wave-uniform
    const int lane = lane_id();
    const int4 wi = reinterpret_cast<const int4 *>(a.win_indices)[w];  // [b,wz,wy,wx]
    const slot_t *tab = a.table + (size_t)wi.x * a.hash_size;
    const int cx = wi.w * a.x_ws + a.x_ws / 2;  // ref :219-225
    const int cy = wi.z * a.y_ws + a.y_ws / 2;
    const int cz = wi.y * a.z_ws + a.z_ws / 2;
    const int e0 = a.n_odd, e1 = e0 + a.n_even, e2 = e1 + a.n_win1, total = e2 + a.n_win2;
    int cnt_odd = 0, cnt_even = 0, cnt_w1 = 0, cnt_w2 = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = 0; base < total; base += MSSVT_WAVE) {
        // every open list full -> nothing can be appended any more
        if (cnt_w2 >= a.max_win2 && cnt_w1 >= a.max_win1 &&
            (cnt_even >= a.max_even || base >= e1) && (cnt_odd >= a.max_odd || base >= e0))
            break;
        const int q = base + lane;
        int seg = (q >= e0) + (q >= e1) + (q >= e2);
        int sv = MSSVT_EMPTY, ox = 0, oy = 0, oz = 0;
        if (q < total) {
            const int *src = seg == 0 ? a.q_odd + q * 3
                           : seg == 1 ? a.q_even + (q - e0) * 3
                           : seg == 2 ? a.q_win1 + (q - e1) * 3
                                      : a.q_win2 + (q - e2) * 3;
            ox = src[0];
            oy = src[1];
            oz = src[2];
            const int sx = cx + ox, sy = cy + oy, sz = cz + oz;
            if (!(sx >= a.x_max || sx < 0 || sy >= a.y_max || sy < 0 || sz >= a.z_max || sz < 0)) {
                const int skey = sx * a.y_max * a.z_max + sy * a.z_max + sz;
                sv = table_find(skey, a.hash_size, tab);
            }
        }
        const bool hit = sv != MSSVT_EMPTY;
        const unsigned long long m_all = __ballot(hit);
        if (m_all == 0) continue;
        const unsigned long long m_odd = __ballot(hit && seg == 0);
        const unsigned long long m_even = __ballot(hit && seg == 1);
        const unsigned long long m_w1 = __ballot(hit && seg <= 2);
        if (hit) {
            if (seg == 0)
                put(a.ind_odd, a.c_odd, w, a.max_odd, cnt_odd + __popcll(m_odd & below), sv, ox, oy, oz);
            if (seg == 1)
                put(a.ind_even, a.c_even, w, a.max_even, cnt_even + __popcll(m_even & below), sv, ox, oy, oz);
            if (seg <= 2)
                put(a.ind_win1, a.c_win1, w, a.max_win1, cnt_w1 + __popcll(m_w1 & below), sv, ox, oy, oz);
            put(a.ind_win2, a.c_win2, w, a.max_win2, cnt_w2 + __popcll(m_all & below), sv, ox, oy, oz);
        }
        cnt_odd += __popcll(m_odd);
        cnt_even += __popcll(m_even);
        cnt_w1 += __popcll(m_w1);
        cnt_w2 += __popcll(m_all);
    }
}

__global__ void __launch_bounds__(WAVES_PER_BLOCK *MSSVT_WAVE)
    k_gather_one_window(int x_max, int y_max, int z_max, int x_ws, int y_ws, int z_ws, int max_win1,
                        int num_wins, int hash_size, int n_win1, int *ind_win1, int *c_win1,
                        const int *q_win1, const int *win_indices, const slot_t *table) {
    const int w = blockIdx.x * WAVES_PER_BLOCK + threadIdx.x / MSSVT_WAVE;
    if (w >= num_wins) return;
    const int lane = lane_id();
    const int4 wi = reinterpret_cast<const int4 *>(win_indices)[w];
    const slot_t *tab = table + (size_t)wi.x * hash_size;
    const int cx = wi.w * x_ws + x_ws / 2;
    const int cy = wi.z * y_ws + y_ws / 2;
    const int cz = wi.y * z_ws + z_ws / 2;
    int cnt = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = 0; base < n_win1 && cnt < max_win1; base += MSSVT_WAVE) {
        const int q = base + lane;
        int sv = MSSVT_EMPTY, ox = 0, oy = 0, oz = 0;
        if (q < n_win1) {
            ox = q_win1[q * 3 + 0];
            oy = q_win1[q * 3 + 1];
            oz = q_win1[q * 3 + 2];
            const int sx = cx + ox, sy = cy + oy, sz = cz + oz;
            if (!(sx >= x_max || sx < 0 || sy >= y_max || sy < 0 || sz >= z_max || sz < 0))
                sv = table_find(sx * y_max * z_max + sy * z_max + sz, hash_size, tab);
        }
        const bool hit = sv != MSSVT_EMPTY;
        const unsigned long long m = __ballot(hit);
        if (hit) put(ind_win1, c_win1, w, max_win1, cnt + __popcll(m & below), sv, ox, oy, oz);
        cnt += __popcll(m);
    }
}

extern "C" int mssvt_gather_two_window_voxels_with_hash(
    int x_max, int y_max, int z_max, int x_ws, int y_ws, int z_ws, int max_num_odd,
    int max_num_even, int max_num_win1, int max_num_win2, int num_wins, int hash_size, int num_odd,
    int num_even, int num_win1, int num_win2, int *vox_ind_odd, int *vox_ind_even,
    int *vox_ind_win1, int *vox_ind_win2, int *vox_coord_odd, int *vox_coord_even,
    int *vox_coord_win1, int *vox_coord_win2, const int *vox_query_odd, const int *vox_query_even,
    const int *vox_query_win1, const int *vox_query_win2, const int *win_indices,
    const int *xyz_to_vidx, void *stream) {
    if (num_wins < 0 || hash_size <= 0 || !xyz_to_vidx) return MSSVT_E_BADARG;
    if (num_wins == 0) return MSSVT_OK;
    if (!win_indices || !vox_ind_odd || !vox_ind_even || !vox_ind_win1 || !vox_ind_win2 ||
        !vox_coord_odd || !vox_coord_even || !vox_coord_win1 || !vox_coord_win2)
        return MSSVT_E_BADARG;
    TwoWinArgs a{x_max, y_max, z_max, x_ws, y_ws, z_ws, max_num_odd, max_num_even, max_num_win1,
                 max_num_win2, num_wins, hash_size, num_odd, num_even, num_win1, num_win2,
                 vox_ind_odd, vox_ind_even, vox_ind_win1, vox_ind_win2, vox_coord_odd,
                 vox_coord_even, vox_coord_win1, vox_coord_win2, vox_query_odd, vox_query_even,
                 vox_query_win1, vox_query_win2, win_indices,
                 reinterpret_cast<const slot_t *>(xyz_to_vidx)};
    k_gather_two_window<<<divup(num_wins, WAVES_PER_BLOCK), WAVES_PER_BLOCK * MSSVT_WAVE, 0,
                          (hipStream_t)stream>>>(a);
    return mssvt_launch_status();
}

extern "C" int mssvt_gather_one_window_voxels_with_hash(
    int x_max, int y_max, int z_max, int x_ws, int y_ws, int z_ws, int max_num_win1, int num_wins,
    int hash_size, int num_win1, int *vox_ind_win1, int *vox_coord_win1, const int *vox_query_win1,
    const int *win_indices, const int *xyz_to_vidx, void *stream) {
    if (num_wins < 0 || hash_size <= 0 || !xyz_to_vidx) return MSSVT_E_BADARG;
    if (num_wins == 0) return MSSVT_OK;
    if (!win_indices || !vox_ind_win1 || !vox_coord_win1 || !vox_query_win1) return MSSVT_E_BADARG;
    k_gather_one_window<<<divup(num_wins, WAVES_PER_BLOCK), WAVES_PER_BLOCK * MSSVT_WAVE, 0,
                          (hipStream_t)stream>>>(
        x_max, y_max, z_max, x_ws, y_ws, z_ws, max_num_win1, num_wins, hash_size, num_win1,
        vox_ind_win1, vox_coord_win1, vox_query_win1, win_indices,
        reinterpret_cast<const slot_t *>(xyz_to_vidx));
    return mssvt_launch_status();
}
