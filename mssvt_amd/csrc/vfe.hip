// vfe.hip -- per-voxel reductions of DynamicVFE without torch_scatter.
//
// The reference reduces point features into voxels with the un-vendored torch_scatter package
// (ref: pcdet/models/backbones_3d/vfe/dynamic_vfe.py:4-8,98,111,128-129):
//     xyz_mean = scatter_mean(xyz, unq_inv)            -> cluster centre of every voxel
//     fea_v    = scatter_max(points_fea, unq_inv)[0]   -> per-voxel max of the PFN activations
// Both are order independent here, hence run-to-run deterministic (torch_scatter's float atomicAdd mean
// is not): the mean accumulates 64-bit FIXED-POINT sums (2^-20 m resolution, exact integer adds), the max
// uses the order-preserving integer view of IEEE floats.  point_voxel (P) = unq_inv with -1 for points
// outside the grid (mssvt_voxelize).
#include "common.hip.h"

#define VFE_FIX 1048576.0  // 2^20 steps per metre

__global__ void __launch_bounds__(256)
    k_vfe_sum_xyz(const float *points, int stride, long long n, const int *point_voxel, long long *sum3, int *cnt) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int v = point_voxel[i];
    if (v < 0) return;
    const float *p = points + i * stride;  // [b, x, y, z, ...]
#pragma unroll
    for (int k = 0; k < 3; ++k)
        atomicAdd(reinterpret_cast<unsigned long long *>(sum3 + (size_t)v * 3 + k),
                  (unsigned long long)__double2ll_rn((double)p[1 + k] * VFE_FIX));
    atomicAdd(cnt + v, 1);
}

__global__ void __launch_bounds__(256) k_vfe_mean_xyz(const long long *sum3, const int *cnt, int num_voxels, float *mean3) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= num_voxels * 3) return;
    const int c = cnt[e / 3];
    mean3[e] = c > 0 ? (float)((double)sum3[e] / VFE_FIX / (double)c) : 0.0f;
}

extern "C" int mssvt_voxel_mean_xyz(const float *points, int point_stride, long long num_points,
                                    const int *point_voxel, int num_voxels, float *mean3, int *count,
                                    long long *scratch_sum3, void *stream_) {
    if ((!points && num_points > 0) || (!point_voxel && num_points > 0) || point_stride < 4 || num_points < 0 ||
        num_voxels < 0 || !mean3 || !count || !scratch_sum3)
        return MSSVT_E_BADARG;
    if (num_voxels == 0) return MSSVT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(scratch_sum3, 0, (size_t)num_voxels * 3 * sizeof(long long), stream);
    if (e == hipSuccess) e = hipMemsetAsync(count, 0, (size_t)num_voxels * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    if (num_points > 0)
        k_vfe_sum_xyz<<<divup(num_points, 256), 256, 0, stream>>>(points, point_stride, num_points, point_voxel,
                                                                  scratch_sum3, count);
    k_vfe_mean_xyz<<<divup((long long)num_voxels * 3, 256), 256, 0, stream>>>(scratch_sum3, count, num_voxels, mean3);
    return mssvt_launch_status();
}

// float max through integer atomics: non-negative floats order like signed ints, negative floats in
// reverse like unsigned ints
__device__ __forceinline__ void atomic_max_float(float *addr, float v) {
    if (v >= 0.0f)
        atomicMax(reinterpret_cast<int *>(addr), __builtin_bit_cast(int, v));
    else
        atomicMin(reinterpret_cast<unsigned int *>(addr), __builtin_bit_cast(unsigned int, v));
}

__global__ void __launch_bounds__(256) k_vfe_fill(float *out, long long n, float value) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = value;
}

// one wavefront per point, lanes over channels: coalesced row reads, atomics spread over the voxel's row
__global__ void __launch_bounds__(256)
    k_vfe_max(const float *feat, int F, long long n, const int *point_voxel, float *out) {
    const long long i = (long long)blockIdx.x * 4 + threadIdx.x / MSSVT_WAVE;
    if (i >= n) return;
    const int v = point_voxel[i];
    if (v < 0) return;
    for (int c = lane_id(); c < F; c += MSSVT_WAVE) atomic_max_float(out + (size_t)v * F + c, feat[i * F + c]);
}

extern "C" int mssvt_voxel_max(const float *features, int F, long long num_points, const int *point_voxel,
                               int num_voxels, float *out, void *stream_) {
    if ((!features && num_points > 0) || (!point_voxel && num_points > 0) || F <= 0 || num_points < 0 || num_voxels < 0 ||
        !out)
        return MSSVT_E_BADARG;
    if (num_voxels == 0) return MSSVT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const long long total = (long long)num_voxels * F;
    k_vfe_fill<<<divup(total, 256), 256, 0, stream>>>(out, total, -INFINITY);
    if (num_points > 0) k_vfe_max<<<divup(num_points, 4), 256, 0, stream>>>(features, F, num_points, point_voxel, out);
    return mssvt_launch_status();
}
