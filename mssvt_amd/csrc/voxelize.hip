// voxelize.hip -- points -> sorted unique voxel coordinates + point->voxel map.
//
// Index part of the reference's DynamicVFE.forward
// (ref: pcdet/models/backbones_3d/vfe/dynamic_vfe.py:83-93,114-118):
//     coord = floor((xyz - range_min) / voxel_size)        (float division, then floor)
//     keep 0 <= coord < grid
//     key   = ((b * X + x) * Y + y) * Z + z
//     unq, unq_inv = torch.unique(key, return_inverse=True)   (sorted)
//     voxel_coords = [b, z, y, x] of unq
// The reference gets there with a device radix sort inside torch.unique.  The key
// space is a dense grid (B * X * Y * Z cells: 7.07 M per Waymo scene), so no sort is
// needed on MI355X: an occupancy BITMAP (1 bit per cell, 0.9 MB per scene, L2 resident)
// is filled with atomicOr, a scan of the per-word popcounts turns it into a rank
// table, and the rank of a cell IS its position in the sorted unique list:
//     k_vox_mark   per point : cell -> atomicOr(bitmap)                 (16 B in / point)
//     k_vox_scan   per 1024 words: popcount + block scan (+ carry from a 2nd level)
//     k_vox_emit   per occupied bit: voxel_coords[rank] = (b, z, y, x)
//     k_vox_inv    per point : unq_inv = rank(cell)  (-1 for points outside the grid)
// Results are bit-identical to the sorted-unique formulation (integer work only; the
// one float op is the reference's own (x - min) / vs followed by floor).
#include "common.hip.h"

#define VX_TPB 256
#define VX_SCAN_WORDS 1024  // words per scan block

__device__ __forceinline__ bool point_cell(const float *p, int stride, long long i, float minx, float miny,
                                           float minz, float vsx, float vsy, float vsz, int X, int Y, int Z,
                                           int B, long long &cell) {
    const float *q = p + i * stride;  // [b, x, y, z, ...]
    const int b = (int)q[0];
    const float fx = floorf(__fdiv_rn(__fsub_rn(q[1], minx), vsx));
    const float fy = floorf(__fdiv_rn(__fsub_rn(q[2], miny), vsy));
    const float fz = floorf(__fdiv_rn(__fsub_rn(q[3], minz), vsz));
    if (!(fx >= 0.f && fx < (float)X && fy >= 0.f && fy < (float)Y && fz >= 0.f && fz < (float)Z)) return false;
    if (b < 0 || b >= B) return false;
    cell = (((long long)b * X + (int)fx) * Y + (int)fy) * Z + (int)fz;
    return true;
}

__global__ void __launch_bounds__(VX_TPB)
    k_vox_mark(const float *points, int stride, long long n, float minx, float miny, float minz, float vsx,
               float vsy, float vsz, int X, int Y, int Z, int B, unsigned int *bitmap) {
    const long long i = (long long)blockIdx.x * VX_TPB + threadIdx.x;
    if (i >= n) return;
    long long cell;
    // (a plain read of the word first, to skip the atomic when the bit already shows -- the columns next to the sensor take
    // hundreds of points each -- was measured: 15.5 -> 46 us; the fire-and-forget atomic does not wait, the read does)
    if (point_cell(points, stride, i, minx, miny, minz, vsx, vsy, vsz, X, Y, Z, B, cell))
        atomicOr(bitmap + (cell >> 5), 1u << (cell & 31));
}

// level 1: per block of VX_SCAN_WORDS words, exclusive prefix of popcounts inside the block
// (-> word_rank) and the block total (-> block_sum)
__global__ void __launch_bounds__(VX_TPB)
    k_vox_scan1(const unsigned int *bitmap, long long nwords, int *word_rank, int *block_sum) {
    __shared__ int wsum[VX_TPB / MSSVT_WAVE];
    const long long base = (long long)blockIdx.x * VX_SCAN_WORDS + threadIdx.x * 4;
    int c[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k] = base + k < nwords ? __popc(bitmap[base + k]) : 0;
        tot += c[k];
    }
    // wave inclusive scan of `tot`
    int incl = tot;
    const int lane = lane_id(), wv = threadIdx.x / MSSVT_WAVE;
    for (int off = 1; off < MSSVT_WAVE; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == MSSVT_WAVE - 1) wsum[wv] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wv; ++w) before += wsum[w];
    int run = before + incl - tot;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < nwords) word_rank[base + k] = run;
        run += c[k];
    }
    if (threadIdx.x == VX_TPB - 1) block_sum[blockIdx.x] = before + incl;
}

// level 2 (one workgroup): exclusive scan of the block sums; total -> counters[0]
__global__ void __launch_bounds__(1024) k_vox_scan2(int *block_sum, int nblocks, int *counters) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int idx = base + threadIdx.x;
        const int v = idx < nblocks ? block_sum[idx] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += t;
            __syncthreads();
        }
        const int incl = part[threadIdx.x];
        if (idx < nblocks) block_sum[idx] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) counters[0] = carry;
}

__global__ void __launch_bounds__(VX_TPB)
    k_vox_emit(const unsigned int *bitmap, long long nwords, const int *word_rank, const int *block_sum, int X,
               int Y, int Z, int capacity, int *voxel_coords) {
    const long long wi = (long long)blockIdx.x * VX_TPB + threadIdx.x;
    if (wi >= nwords) return;
    unsigned int bits = bitmap[wi];
    if (!bits) return;
    int rank = word_rank[wi] + block_sum[wi / VX_SCAN_WORDS];
    while (bits) {
        const int bit = __ffs(bits) - 1;
        bits &= bits - 1;
        const long long cell = wi * 32 + bit;
        const int z = (int)(cell % Z);
        const int y = (int)((cell / Z) % Y);
        const int x = (int)((cell / ((long long)Z * Y)) % X);
        const int b = (int)(cell / ((long long)Z * Y * X));
        if (rank < capacity) reinterpret_cast<int4 *>(voxel_coords)[rank] = make_int4(b, z, y, x);
        ++rank;
    }
}

__global__ void __launch_bounds__(VX_TPB)
    k_vox_inv(const float *points, int stride, long long n, float minx, float miny, float minz, float vsx,
              float vsy, float vsz, int X, int Y, int Z, int B, const unsigned int *bitmap, const int *word_rank,
              const int *block_sum, int *point_voxel) {
    const long long i = (long long)blockIdx.x * VX_TPB + threadIdx.x;
    if (i >= n) return;
    long long cell;
    int r = -1;
    if (point_cell(points, stride, i, minx, miny, minz, vsx, vsy, vsz, X, Y, Z, B, cell)) {
        const long long wi = cell >> 5;
        r = word_rank[wi] + block_sum[wi / VX_SCAN_WORDS] + __popc(bitmap[wi] & ((1u << (cell & 31)) - 1u));
    }
    point_voxel[i] = r;
}

extern "C" long long mssvt_voxelize_workspace_ints(int batch_size, int X, int Y, int Z) {
    const long long cells = (long long)batch_size * X * Y * Z;
    const long long nwords = (cells + 31) / 32;
    const long long nblocks = (nwords + VX_SCAN_WORDS - 1) / VX_SCAN_WORDS;
    return 2 * nwords + nblocks + 16;  // bitmap | word_rank | block_sum | counters
}

extern "C" int mssvt_voxelize(const float *points, int point_stride, long long num_points, int batch_size,
                              const float *host_range_min3, const float *host_voxel_size3, int X, int Y,
                              int Z, int voxel_capacity, int *voxel_coords, int *point_voxel,
                              int *num_voxels_dev, int *workspace, void *stream_) {
    if (!points || point_stride < 4 || num_points < 0 || batch_size <= 0 || !host_range_min3 ||
        !host_voxel_size3 || X <= 0 || Y <= 0 || Z <= 0 || !voxel_coords || !num_voxels_dev || !workspace)
        return MSSVT_E_BADARG;
    hipStream_t stream = (hipStream_t)stream_;
    const long long cells = (long long)batch_size * X * Y * Z;
    const long long nwords = (cells + 31) / 32;
    const long long nblocks = (nwords + VX_SCAN_WORDS - 1) / VX_SCAN_WORDS;
    if (nblocks > 0x7FFFFFFF || num_points / VX_TPB > 0x7FFFFFF0LL) return MSSVT_E_TOOLARGE;
    unsigned int *bitmap = reinterpret_cast<unsigned int *>(workspace);
    int *word_rank = workspace + nwords;
    int *block_sum = word_rank + nwords;
    hipError_t e = hipMemsetAsync(bitmap, 0, nwords * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    const float mx = host_range_min3[0], my = host_range_min3[1], mz = host_range_min3[2];
    const float sx = host_voxel_size3[0], sy = host_voxel_size3[1], sz = host_voxel_size3[2];
    if (num_points > 0)
        k_vox_mark<<<divup(num_points, VX_TPB), VX_TPB, 0, stream>>>(points, point_stride, num_points, mx, my, mz,
                                                                    sx, sy, sz, X, Y, Z, batch_size, bitmap);
    k_vox_scan1<<<(int)nblocks, VX_TPB, 0, stream>>>(bitmap, nwords, word_rank, block_sum);
    k_vox_scan2<<<1, 1024, 0, stream>>>(block_sum, (int)nblocks, num_voxels_dev);
    k_vox_emit<<<divup(nwords, VX_TPB), VX_TPB, 0, stream>>>(bitmap, nwords, word_rank, block_sum, X, Y, Z,
                                                            voxel_capacity, voxel_coords);
    if (point_voxel && num_points > 0)
        k_vox_inv<<<divup(num_points, VX_TPB), VX_TPB, 0, stream>>>(points, point_stride, num_points, mx, my, mz,
                                                                   sx, sy, sz, X, Y, Z, batch_size, bitmap,
                                                                   word_rank, block_sum, point_voxel);
    return mssvt_launch_status();
}
