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
// Several partitions of the SAME voxel list (a Block's windows and the following CompressBlock's)
// run in the same five launches: blockIdx.y = partition.  Per-set workspace layout:
// [WS_HDR_INTS header | sample_base (B+1) | flags (n) | blockcnt].
#define WP_MAX_SETS 4
struct WinSet {
    WinKey kf;
    int num_windows;
    slot_t *table;    // final table (key -> window rank)
    slot_t *phase1;   // table of the min-insert pass: a caller-provided scratch table or `table` itself
    int *w_indices, *win_compact, *vcount, *ws;
};
struct WinSets {
    WinSet s[WP_MAX_SETS];
};
__device__ __forceinline__ int *ws_sample_base(int *ws) { return ws + WS_HDR_INTS; }
__device__ __forceinline__ int *ws_flags(int *ws, int batch_size) { return ws + WS_HDR_INTS + batch_size + 1; }
__device__ __forceinline__ int *ws_blockcnt(int *ws, int batch_size, int n) { return ws_flags(ws, batch_size) + n; }

// pass 1: order-agnostic insert, value = min voxel index of the window
__global__ void __launch_bounds__(TPB) k_win_insert_min(WinSets sets, int n, int hash_size,
                                                        int batch_size, const int *v_indices) {
    const WinSet &S = sets.s[blockIdx.y];
    int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    int b, key, wz, wy, wx;
    if (!S.kf(v_indices, i, b, key, wz, wy, wx) || b < 0 || b >= batch_size) return;
    if (table_insert_min(key, i, hash_size, S.phase1 + (size_t)b * hash_size) < 0)
        atomicOr(S.ws + WS_STATUS, ST_TABLE_OVERFLOW);
}

// pass 2: flag first occurrences, count them per 256-voxel block
__global__ void __launch_bounds__(TPB) k_win_flag_count(WinSets sets, int n, int hash_size,
                                                        int batch_size, const int *v_indices) {
    const WinSet &S = sets.s[blockIdx.y];
    int *flags = ws_flags(S.ws, batch_size), *blockcnt = ws_blockcnt(S.ws, batch_size, n);
    __shared__ int wcnt[TPB / MSSVT_WAVE];
    int i = blockIdx.x * TPB + threadIdx.x;
    int flag = 0;
    if (i < n) {
        int b, key, wz, wy, wx;
        if (S.kf(v_indices, i, b, key, wz, wy, wx) && b >= 0 && b < batch_size)
            flag = table_find(key, hash_size, S.phase1 + (size_t)b * hash_size) == i;
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

// pass 3 (one workgroup per partition): exclusive scan of the block counts, per-sample bases.
// Samples are contiguous in v_indices (as every consumer of the reference
// assumes, e.g. ref :81-87), so sample b starts at lower_bound(batch >= b).
__global__ void __launch_bounds__(1024) k_win_scan(WinSets sets, int n, int nblocks, int batch_size,
                                                   const int *v_indices) {
    const WinSet &S = sets.s[blockIdx.y];
    int *ws = S.ws, *sample_base = ws_sample_base(ws), *vcount = S.vcount;
    const int *flags = ws_flags(ws, batch_size);
    int *blockcnt = ws_blockcnt(ws, batch_size, n);
    const int num_windows = S.num_windows;
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
__global__ void __launch_bounds__(TPB) k_win_insert_ranked(WinSets sets, int n, int hash_size,
                                                           int batch_size, const int *v_indices) {
    const WinSet &S = sets.s[blockIdx.y];
    int *ws = S.ws;
    const int *flags = ws_flags(ws, batch_size), *blockoff = ws_blockcnt(ws, batch_size, n);
    const int *sample_base = ws_sample_base(ws);
    const int num_windows = S.num_windows;
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
    S.kf(v_indices, i, b, key, wz, wy, wx);
    int rank = r - sample_base[b];
    if (S.win_compact) {  // (nw,4) rows [b,wz,wy,wx]: ref mssvt/mssvt_ops.py:45-53 done on device
        reinterpret_cast<int4 *>(S.win_compact)[r] = make_int4(b, wz, wy, wx);
    }
    if (rank >= num_windows) return;  // the reference writes out of bounds here
    if (S.w_indices) {
        int *w = S.w_indices + ((size_t)b * num_windows + rank) * 3;
        w[0] = wz;  // ref :154-156
        w[1] = wy;
        w[2] = wx;
    }
    int st = table_insert_ordered(key, rank, hash_size, S.table + (size_t)b * hash_size);
    if (st & ST_TABLE_OVERFLOW) atomicOr(ws + WS_STATUS, ST_TABLE_OVERFLOW);
}

// refill of the tables that served as their own phase-1 scratch (blockIdx.y = partition)
__global__ void k_fill_slots_sets(WinSets sets, long long n) {
    const WinSet &S = sets.s[blockIdx.y];
    if (S.phase1 != S.table) return;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) S.table[i] = SLOT_EMPTY;
}

// ---- host entry points --------------------------------------------------------
extern "C" long long mssvt_hash_workspace_ints(int num_voxels, int batch_size) {
    long long n = num_voxels > 0 ? num_voxels : 0;
    return WS_HDR_INTS + (batch_size + 1) + n + (n + TPB - 1) / TPB + 64;
}

static int build_mapping_launch(int x_max, int y_max, int z_max, int num_voxels, int hash_size, int batch_size,
                                const int *v_indices, const int *v_bs_cnt, int *xyz_to_vidx, int *workspace,
                                hipStream_t stream);

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
    return build_mapping_launch(x_max, y_max, z_max, num_voxels, hash_size, batch_size, v_indices, v_bs_cnt,
                                xyz_to_vidx, workspace, stream);
}

static int build_mapping_launch(int x_max, int y_max, int z_max, int num_voxels, int hash_size, int batch_size,
                                const int *v_indices, const int *v_bs_cnt, int *xyz_to_vidx, int *workspace,
                                hipStream_t stream) {
    if (num_voxels == 0) return MSSVT_OK;  // an empty table (the caller pre-fills it with -1)
    VoxKey kf{x_max, y_max, z_max};
    slot_t *table = reinterpret_cast<slot_t *>(xyz_to_vidx);
    k_vox_insert_all<<<divup(num_voxels, TPB), TPB, 0, stream>>>(
        kf, num_voxels, hash_size, batch_size, v_indices, v_bs_cnt, table, workspace);
    k_vox_dup_fallback<<<1, 1024, 0, stream>>>(kf, num_voxels, hash_size, batch_size, v_indices,
                                               v_bs_cnt, table, workspace);
    return mssvt_launch_status();
}

// all partitions of `sets` over one voxel list, five launches (grid.y = partition)
static int window_partition_sets(const WinSets &sets, int num_sets, bool refill, int num_voxels, int hash_size,
                                 int batch_size, const int *v_indices, hipStream_t stream) {
    const int nblocks = divup(num_voxels, TPB);
    const long long cells = (long long)batch_size * hash_size;
    k_win_insert_min<<<dim3(nblocks, num_sets), TPB, 0, stream>>>(sets, num_voxels, hash_size, batch_size, v_indices);
    k_win_flag_count<<<dim3(nblocks, num_sets), TPB, 0, stream>>>(sets, num_voxels, hash_size, batch_size, v_indices);
    k_win_scan<<<dim3(1, num_sets), 1024, 0, stream>>>(sets, num_voxels, nblocks, batch_size, v_indices);
    if (refill) {
        long long fill_blocks = (cells + 1023) / 1024;
        if (fill_blocks > 2048) fill_blocks = 2048;  // grid-stride the rest
        k_fill_slots_sets<<<dim3((int)fill_blocks, num_sets), 1024, 0, stream>>>(sets, cells);
    }
    k_win_insert_ranked<<<dim3(nblocks, num_sets), TPB, 0, stream>>>(sets, num_voxels, hash_size, batch_size, v_indices);
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
    WinSets sets;
    WinSet &S = sets.s[0];
    S.kf = WinKey{x_wgs, y_wgs, z_wgs, x_ws, y_ws, z_ws};
    S.num_windows = num_windows;
    S.table = S.phase1 = reinterpret_cast<slot_t *>(xyz_to_vidx);
    S.w_indices = w_indices;
    S.win_compact = win_compact;
    S.vcount = vcount;
    S.ws = workspace;
    return window_partition_sets(sets, 1, true, num_voxels, hash_size, batch_size, v_indices, stream);
}

// host arrays of mssvt_window_partition_multi -> kernel argument pack
static int window_sets_from_host(WinSets &sets, bool &refill, int num_sets, const int *host_win_grid3,
                                 const int *host_win_size3, const int *host_max_num_wins, int num_voxels,
                                 int hash_size, int batch_size, const int *v_indices, int *const *host_win_ind,
                                 int *const *host_tables, int *const *host_scratch_tables, int *const *host_vcount,
                                 int *workspaces, long long workspace_stride_ints) {
    if (num_sets < 1 || num_sets > WP_MAX_SETS || !host_win_grid3 || !host_win_size3 || !host_max_num_wins ||
        !host_win_ind || !host_tables || !host_vcount || !workspaces || (!v_indices && num_voxels > 0) ||
        hash_size <= 0 || batch_size <= 0 || num_voxels < 0 ||
        workspace_stride_ints < mssvt_hash_workspace_ints(num_voxels, batch_size))
        return MSSVT_E_BADARG;
    refill = false;
    for (int k = 0; k < num_sets; ++k) {
        const int *g = host_win_grid3 + 3 * k, *w = host_win_size3 + 3 * k;
        if (!host_win_ind[k] || !host_tables[k] || !host_vcount[k] || w[0] <= 0 || w[1] <= 0 || w[2] <= 0)
            return MSSVT_E_BADARG;
        WinSet &S = sets.s[k];
        S.kf = WinKey{g[0], g[1], g[2], w[0], w[1], w[2]};
        S.num_windows = host_max_num_wins[k];
        S.table = reinterpret_cast<slot_t *>(host_tables[k]);
        S.phase1 = host_scratch_tables && host_scratch_tables[k] ? reinterpret_cast<slot_t *>(host_scratch_tables[k])
                                                                 : S.table;
        refill = refill || S.phase1 == S.table;
        S.w_indices = nullptr;
        S.win_compact = host_win_ind[k];
        S.vcount = host_vcount[k];
        S.ws = workspaces + (size_t)k * workspace_stride_ints;
    }
    return MSSVT_OK;
}

// Several partitions of one voxel list in the same launches (see include/mssvt_hip.h)
extern "C" int mssvt_window_partition_multi(int num_sets, const int *host_win_grid3, const int *host_win_size3,
                                            const int *host_max_num_wins, int num_voxels, int hash_size,
                                            int batch_size, const int *v_indices, int *const *host_win_ind,
                                            int *const *host_tables, int *const *host_scratch_tables,
                                            int *const *host_vcount, int *workspaces, long long workspace_stride_ints,
                                            void *stream_) {
    WinSets sets;
    bool refill = false;
    int rc = window_sets_from_host(sets, refill, num_sets, host_win_grid3, host_win_size3, host_max_num_wins,
                                   num_voxels, hash_size, batch_size, v_indices, host_win_ind, host_tables,
                                   host_scratch_tables, host_vcount, workspaces, workspace_stride_ints);
    if (rc != MSSVT_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    // the headers of all partitions (status, window count) are cleared by ONE fill: the workspaces are
    // slices of one allocation
    // (rounded up to 4 KiB inside the allocation: the runtime splits a fill whose size is not a multiple of
    // 16 bytes into two launches; the words behind a header are scratch that is written before it is read)
    size_t zero_bytes = ((size_t)(num_sets - 1) * workspace_stride_ints + WS_HDR_INTS) * sizeof(int);
    const size_t all_bytes = (size_t)num_sets * workspace_stride_ints * sizeof(int);
    if (((zero_bytes + 4095) & ~(size_t)4095) <= all_bytes) zero_bytes = (zero_bytes + 4095) & ~(size_t)4095;
    hipError_t e = hipMemsetAsync(workspaces, 0, zero_bytes, stream);
    if (e != hipSuccess) return (int)e;
    if (num_voxels == 0) {
        for (int k = 0; k < num_sets; ++k) {
            e = hipMemsetAsync(host_vcount[k], 0, batch_size * sizeof(int), stream);
            if (e != hipSuccess) return (int)e;
        }
        return MSSVT_OK;
    }
    return window_partition_sets(sets, num_sets, refill, num_voxels, hash_size, batch_size, v_indices, stream);
}

// Everything a resolution level needs before its first Block, behind ONE fill (see include/mssvt_hip.h)
extern "C" int mssvt_level_setup(int num_voxels, int batch_size, int x_max, int y_max, int z_max, int hash_size,
                                 const int *v_indices, void *zero_region, long long zero_bytes, int *v_bs_cnt,
                                 int *map_table, int *map_workspace, unsigned long long *occ_columns, int num_sets,
                                 const int *host_win_grid3, const int *host_win_size3,
                                 const int *host_max_num_wins, int *const *host_win_ind, int *const *host_tables,
                                 int *const *host_scratch_tables, int *const *host_vcount, int *workspaces,
                                 long long workspace_stride_ints, void *stream_) {
    if (!zero_region || zero_bytes <= 0 || !v_bs_cnt || !map_table || !map_workspace || batch_size <= 0 ||
        hash_size <= 0 || num_voxels < 0 || (!v_indices && num_voxels > 0) || x_max <= 0 || y_max <= 0 || z_max <= 0 ||
        num_sets < 0)
        return MSSVT_E_BADARG;
    if (occ_columns && z_max > 64) return MSSVT_E_TOOLARGE;
    WinSets sets;
    bool refill = false;
    if (num_sets > 0) {
        int rc = window_sets_from_host(sets, refill, num_sets, host_win_grid3, host_win_size3, host_max_num_wins,
                                       num_voxels, hash_size, batch_size, v_indices, host_win_ind, host_tables,
                                       host_scratch_tables, host_vcount, workspaces, workspace_stride_ints);
        if (rc != MSSVT_OK) return rc;
    }
    // every word the kernels below accumulate into must lie inside the region cleared here
    const char *z0 = (const char *)zero_region, *z1 = z0 + zero_bytes;
    auto inside = [&](const void *p, size_t bytes) { return (const char *)p >= z0 && (const char *)p + bytes <= z1; };
    bool ok = inside(v_bs_cnt, (size_t)batch_size * sizeof(int)) && inside(map_workspace, WS_HDR_INTS * sizeof(int)) &&
              (!occ_columns || inside(occ_columns, (size_t)batch_size * x_max * y_max * sizeof(unsigned long long)));
    for (int k = 0; k < num_sets; ++k) ok = ok && inside(sets.s[k].ws, WS_HDR_INTS * sizeof(int));
    if (num_voxels == 0)
        for (int k = 0; k < num_sets; ++k) ok = ok && inside(host_vcount[k], (size_t)batch_size * sizeof(int));
    if (!ok) return MSSVT_E_BADARG;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(zero_region, 0, (size_t)zero_bytes, stream);
    if (e != hipSuccess) return (int)e;
    int rc = mssvt_batch_counts_launch(v_indices, num_voxels, batch_size, v_bs_cnt, stream);
    if (rc != MSSVT_OK) return rc;
    rc = build_mapping_launch(x_max, y_max, z_max, num_voxels, hash_size, batch_size, v_indices, v_bs_cnt, map_table,
                              map_workspace, stream);
    if (rc != MSSVT_OK) return rc;
    if (occ_columns) {
        rc = mssvt_occupancy_columns_launch(v_indices, num_voxels, batch_size, x_max, y_max, z_max, occ_columns, stream);
        if (rc != MSSVT_OK) return rc;
    }
    if (num_sets > 0 && num_voxels > 0)
        return window_partition_sets(sets, num_sets, refill, num_voxels, hash_size, batch_size, v_indices, stream);
    return MSSVT_OK;
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
