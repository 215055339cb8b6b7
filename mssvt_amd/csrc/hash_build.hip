// hash_build.hip -- K1 (voxel hash map) and K2 (non-empty window discovery).
//
// Replaces build_mapping_with_hash_kernel / window_with_hash_kernel
// (ref: mssvt/src/ms_sparse_attention_gpu.cu:66-97, :117-168).  The reference
// lets thousands of threads race through atomicCAS, so its table layout and its
// window numbering depend on thread timing.  Here both are DETERMINISTIC and
// equal to what a sequential pass over the voxels in index order produces
// (= the oracle's canonical orders (a), (b), (c)), while still running fully
// parallel:
//   * distinct keys are placed with table_insert_ordered() (common.hip.h), a
//     priority-ordered linear-probing insert whose quiescent layout is unique;
//   * duplicate keys (the normal case for windows: ~4 voxels per window) are
//     first reduced to their first occurrence with an order-agnostic min-insert,
//     a block-wide ballot scan ranks the first occurrences, and only those are
//     inserted in order.
// All work is int32 / byte traffic -> HBM/L2 latency bound; one thread per voxel,
// 256-thread workgroups (4 waves), no LDS tiles needed.
#include "common.hip.h"

#define TPB 256

// ---- key functors -----------------------------------------------------------
struct VoxKey {  // ref :76-93
    int x_max, y_max, z_max;
    __device__ __forceinline__ bool operator()(const int *vi, int i, int &b, int &key) const {
        const int4 v = reinterpret_cast<const int4 *>(vi)[i];  // [b,z,y,x], 16 B coalesced
        b = v.x;
        int z = v.y, y = v.z, x = v.w;
        if (x >= x_max || x < 0 || y < 0 || y >= y_max || z < 0 || z >= z_max) return false;
        key = x * y_max * z_max + y * z_max + z;
        return true;
    }
};

struct WinKey {  // ref :132-146
    int x_wgs, y_wgs, z_wgs, x_ws, y_ws, z_ws;
    __device__ __forceinline__ bool operator()(const int *vi, int i, int &b, int &key, int &wz,
                                               int &wy, int &wx) const {
        const int4 v = reinterpret_cast<const int4 *>(vi)[i];
        b = v.x;
        wz = v.y / z_ws;
        wy = v.z / y_ws;
        wx = v.w / x_ws;
        if (wx < 0 || wx >= x_wgs || wy < 0 || wy >= y_wgs || wz < 0 || wz >= z_wgs) return false;
        key = wx * y_wgs * z_wgs + wy * z_wgs + wz;
        return true;
    }
};

__device__ __forceinline__ int sample_start(const int *v_bs_cnt, int b) {  // ref :81-86
    int s = 0;
    for (int k = 0; k < b; ++k) s += v_bs_cnt[k];
    return s;
}

// ---- K1 fast path: all voxels are distinct keys ------------------------------
__global__ void __launch_bounds__(TPB) k_vox_insert_all(VoxKey kf, int n, int hash_size,
                                                        int batch_size, const int *v_indices,
                                                        const int *v_bs_cnt, slot_t *table,
                                                        int *ws) {
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    int b, key;
    if (!kf(v_indices, i, b, key)) return;
    if (b < 0 || b >= batch_size) return;
    int v_idx = i - sample_start(v_bs_cnt, b);
    int st = table_insert_ordered(key, v_idx, hash_size, table + (size_t)b * hash_size);
    if (st) atomicOr(ws + WS_STATUS, st);
}

// ---- K1 slow path: duplicate voxel coordinates.  No VFE produces them, but the
// oracle defines the outcome (first inserter owns the slot, last writer owns the
// value, also when the table overflows), so it is reproduced literally: the
// tables are cleared and ONE THREAD PER SAMPLE replays the reference's insertion
// loop (ref :22-41) in voxel-index order.  Slow (sequential per sample) by design:
// it only runs for invalid input, and it exits immediately otherwise. ------------
__global__ void __launch_bounds__(1024) k_vox_dup_fallback(VoxKey kf, int n, int hash_size,
                                                           int batch_size, const int *v_indices,
                                                           const int *v_bs_cnt, slot_t *table,
                                                           int *ws) {
    if (!(ws[WS_STATUS] & ST_DUP)) return;
    const long long cells = (long long)batch_size * hash_size;
    for (long long c = threadIdx.x; c < cells; c += 1024) table[c] = SLOT_EMPTY;
    __syncthreads();
    for (int bs = threadIdx.x; bs < batch_size; bs += 1024) {
        int *tab = reinterpret_cast<int *>(table + (size_t)bs * hash_size);
        const int start = sample_start(v_bs_cnt, bs);
        const int end = min(n, start + v_bs_cnt[bs]);
        for (int i = start; i < end; ++i) {
            int b, key;
            if (!kf(v_indices, i, b, key) || b != bs) continue;
            int h = key % hash_size, prob = 0;
            for (;;) {
                const int prev = tab[2 * h];
                if (prev == MSSVT_EMPTY) tab[2 * h] = key;
                if (prev == MSSVT_EMPTY || prev == key) {
                    tab[2 * h + 1] = i - start;
                    break;
                }
                h = h + 1 == hash_size ? 0 : h + 1;
                if (++prob >= hash_size) {
                    atomicOr(ws + WS_STATUS, ST_TABLE_OVERFLOW);
                    break;
                }
            }
        }
    }
}

// ---- K2 ----------------------------------------------------------------------
// pass 1: order-agnostic insert, value = min voxel index of the window
__global__ void __launch_bounds__(TPB) k_win_insert_min(WinKey kf, int n, int hash_size,
                                                        int batch_size, const int *v_indices,
                                                        slot_t *table, int *ws) {
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    int b, key, wz, wy, wx;
    if (!kf(v_indices, i, b, key, wz, wy, wx) || b < 0 || b >= batch_size) return;
    if (table_insert_min(key, i, hash_size, table + (size_t)b * hash_size) < 0)
        atomicOr(ws + WS_STATUS, ST_TABLE_OVERFLOW);
}

// pass 2: flag first occurrences, count them per 256-voxel block
__global__ void __launch_bounds__(TPB) k_win_flag_count(WinKey kf, int n, int hash_size,
                                                        int batch_size, const int *v_indices,
                                                        const slot_t *table, int *flags,
                                                        int *blockcnt) {
    __shared__ int wcnt[TPB / MSSVT_WAVE];
    int i = blockIdx.x * TPB + threadIdx.x;
    int flag = 0;
    if (i < n) {
        int b, key, wz, wy, wx;
        if (kf(v_indices, i, b, key, wz, wy, wx) && b >= 0 && b < batch_size)
            flag = table_find(key, hash_size, table + (size_t)b * hash_size) == i;
        flags[i] = flag;
    }
    unsigned long long m = __ballot(flag);
    if (lane_id() == 0) wcnt[threadIdx.x / MSSVT_WAVE] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < TPB / MSSVT_WAVE; ++w) s += wcnt[w];
        blockcnt[blockIdx.x] = s;
    }
}

// pass 3 (one workgroup): exclusive scan of the block counts, per-sample bases.
// Samples are contiguous in v_indices (as every consumer of the reference
// assumes, e.g. ref :81-87), so sample b starts at lower_bound(batch >= b).
__global__ void __launch_bounds__(1024) k_win_scan(int n, int nblocks, int batch_size,
                                                   int num_windows, const int *v_indices,
                                                   const int *flags, int *blockcnt,
                                                   int *sample_base, int *vcount, int *ws) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        int idx = base + threadIdx.x;
        int v = idx < nblocks ? blockcnt[idx] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            int t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        int incl = part[threadIdx.x];
        if (idx < nblocks) blockcnt[idx] = carry + incl - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    const int total = carry;
    for (int b = threadIdx.x; b <= batch_size; b += 1024) {
        int start = n;
        if (b < batch_size) {
            int lo = 0, hi = n;  // first i with batch(i) >= b
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (v_indices[mid * 4] >= b) hi = mid; else lo = mid + 1;
            }
            start = lo;
        }
        int r = total;
        if (start < n) {
            int blk = start / TPB;
            r = blockcnt[blk];
            for (int i = blk * TPB; i < start; ++i) r += flags[i];
        }
        sample_base[b] = r;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < batch_size; b += 1024) {
        int c = sample_base[b + 1] - sample_base[b];
        vcount[b] = c;  // ref: vcount after the kernel = windows per sample
        if (c > num_windows) atomicOr(ws + WS_STATUS, ST_WIN_OVERFLOW);
    }
    if (threadIdx.x == 0) ws[1] = total;
}

// pass 4: rank first occurrences, emit window rows, ordered insert (key, rank)
__global__ void __launch_bounds__(TPB) k_win_insert_ranked(WinKey kf, int n, int hash_size,
                                                           int batch_size, int num_windows,
                                                           const int *v_indices, const int *flags,
                                                           const int *blockoff,
                                                           const int *sample_base, slot_t *table,
                                                           int *w_indices, int *win_compact,
                                                           int *ws) {
    __shared__ int woff[TPB / MSSVT_WAVE];
    int i = blockIdx.x * TPB + threadIdx.x;
    int flag = i < n ? flags[i] : 0;
    unsigned long long m = __ballot(flag);
    int lane = lane_id(), wv = threadIdx.x / MSSVT_WAVE;
    if (lane == 0) woff[wv] = __popcll(m);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wv; ++w) before += woff[w];
    if (!flag) return;
    int r = blockoff[blockIdx.x] + before + __popcll(m & ((1ull << lane) - 1ull));
    int b, key, wz, wy, wx;
    kf(v_indices, i, b, key, wz, wy, wx);
    int rank = r - sample_base[b];
    if (win_compact) {  // (nw,4) rows [b,wz,wy,wx]: ref mssvt/mssvt_ops.py:45-53 done on device
        reinterpret_cast<int4 *>(win_compact)[r] = make_int4(b, wz, wy, wx);
    }
    if (rank >= num_windows) return;  // the reference writes out of bounds here
    if (w_indices) {
        int *w = w_indices + ((size_t)b * num_windows + rank) * 3;
        w[0] = wz;  // ref :154-156
        w[1] = wy;
        w[2] = wx;
    }
    int st = table_insert_ordered(key, rank, hash_size, table + (size_t)b * hash_size);
    if (st & ST_TABLE_OVERFLOW) atomicOr(ws + WS_STATUS, ST_TABLE_OVERFLOW);
}

__global__ void k_fill_slots(slot_t *p, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = SLOT_EMPTY;
}

// ---- host entry points --------------------------------------------------------
extern "C" long long mssvt_hash_workspace_ints(int num_voxels, int batch_size) {
    long long n = num_voxels > 0 ? num_voxels : 0;
    return WS_HDR_INTS + (batch_size + 1) + n + (n + TPB - 1) / TPB + 64;
}

extern "C" int mssvt_build_mapping_with_hash(int x_max, int y_max, int z_max, int num_voxels,
                                             int hash_size, int batch_size, const int *v_indices,
                                             const int *v_bs_cnt, int *xyz_to_vidx, int *workspace,
                                             void *stream_) {
    if ((!v_indices && num_voxels > 0) || !v_bs_cnt || !xyz_to_vidx || !workspace || hash_size <= 0 ||
        batch_size <= 0 || num_voxels < 0)
        return MSSVT_E_BADARG;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(workspace, 0, WS_HDR_INTS * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    if (num_voxels == 0) return MSSVT_OK;  // an empty table (the caller pre-fills it with -1)
    VoxKey kf{x_max, y_max, z_max};
    slot_t *table = reinterpret_cast<slot_t *>(xyz_to_vidx);
    k_vox_insert_all<<<divup(num_voxels, TPB), TPB, 0, stream>>>(
        kf, num_voxels, hash_size, batch_size, v_indices, v_bs_cnt, table, workspace);
    k_vox_dup_fallback<<<1, 1024, 0, stream>>>(kf, num_voxels, hash_size, batch_size, v_indices,
                                               v_bs_cnt, table, workspace);
    return mssvt_launch_status();
}

// shared by the reference-shaped entry point and the compact one (fused.hip)
int mssvt_window_partition_impl(int x_wgs, int y_wgs, int z_wgs, int x_ws, int y_ws, int z_ws,
                                int num_voxels, int num_windows, int hash_size, int batch_size,
                                const int *v_indices, int *w_indices, int *win_compact,
                                int *xyz_to_vidx, int *vcount, int *workspace,
                                hipStream_t stream) {
    hipError_t e = hipMemsetAsync(workspace, 0, WS_HDR_INTS * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    if (num_voxels == 0) {
        e = hipMemsetAsync(vcount, 0, batch_size * sizeof(int), stream);
        return e == hipSuccess ? MSSVT_OK : (int)e;
    }
    WinKey kf{x_wgs, y_wgs, z_wgs, x_ws, y_ws, z_ws};
    slot_t *table = reinterpret_cast<slot_t *>(xyz_to_vidx);
    const int nblocks = divup(num_voxels, TPB);
    int *sample_base = workspace + WS_HDR_INTS;
    int *flags = sample_base + (batch_size + 1);
    int *blockcnt = flags + num_voxels;
    const long long cells = (long long)batch_size * hash_size;
    k_win_insert_min<<<nblocks, TPB, 0, stream>>>(kf, num_voxels, hash_size, batch_size,
                                                   v_indices, table, workspace);
    k_win_flag_count<<<nblocks, TPB, 0, stream>>>(kf, num_voxels, hash_size, batch_size,
                                                   v_indices, table, flags, blockcnt);
    k_win_scan<<<1, 1024, 0, stream>>>(num_voxels, nblocks, batch_size, num_windows, v_indices,
                                       flags, blockcnt, sample_base, vcount, workspace);
    long long fill_blocks = (cells + 1023) / 1024;
    if (fill_blocks > 2048) fill_blocks = 2048;  // grid-stride the rest
    k_fill_slots<<<(int)fill_blocks, 1024, 0, stream>>>(table, cells);
    k_win_insert_ranked<<<nblocks, TPB, 0, stream>>>(kf, num_voxels, hash_size, batch_size,
                                                      num_windows, v_indices, flags, blockcnt,
                                                      sample_base, table, w_indices, win_compact,
                                                      workspace);
    return mssvt_launch_status();
}

extern "C" int mssvt_window_with_hash(int x_wgs, int y_wgs, int z_wgs, int x_ws, int y_ws,
                                      int z_ws, int num_voxels, int num_windows, int hash_size,
                                      int batch_size, const int *v_indices, int *w_indices,
                                      int *xyz_to_vidx, int *vcount, int *workspace,
                                      void *stream) {
    if (!v_indices || !w_indices || !xyz_to_vidx || !vcount || !workspace || hash_size <= 0 ||
        batch_size <= 0 || num_voxels < 0 || x_ws <= 0 || y_ws <= 0 || z_ws <= 0)
        return MSSVT_E_BADARG;
    return mssvt_window_partition_impl(x_wgs, y_wgs, z_wgs, x_ws, y_ws, z_ws, num_voxels,
                                       num_windows, hash_size, batch_size, v_indices, w_indices,
                                       nullptr, xyz_to_vidx, vcount, workspace,
                                       (hipStream_t)stream);
}

extern "C" int mssvt_window_partition_compact(int x_wgs, int y_wgs, int z_wgs, int x_ws, int y_ws,
                                              int z_ws, int num_voxels, int max_num_wins,
                                              int hash_size, int batch_size, const int *v_indices,
                                              int *win_ind, int *xyz_to_vidx, int *vcount,
                                              int *workspace, void *stream) {
    if (!v_indices || !win_ind || !xyz_to_vidx || !vcount || !workspace || hash_size <= 0 ||
        batch_size <= 0 || num_voxels < 0 || x_ws <= 0 || y_ws <= 0 || z_ws <= 0)
        return MSSVT_E_BADARG;
    return mssvt_window_partition_impl(x_wgs, y_wgs, z_wgs, x_ws, y_ws, z_ws, num_voxels,
                                       max_num_wins, hash_size, batch_size, v_indices, nullptr,
                                       win_ind, xyz_to_vidx, vcount, workspace,
                                       (hipStream_t)stream);
}
