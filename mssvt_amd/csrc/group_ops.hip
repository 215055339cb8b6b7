// group_ops.hip -- row/point gathers of the granular operator surface:
//   K5  mssvt_group_features        (ref: mssvt/src/group_features_gpu.cu:73-106)
//   K6  mssvt_group_features_grad   (ref: group_features_gpu.cu:15-47)
//   K8  mssvt_gather_points(+grad)  (ref: pointnet2/pointnet2_batch/src/sampling_gpu.cu:15-31, :53-90)
//   K10 mssvt_group_points(+grad)   (ref: pointnet2/pointnet2_batch/src/group_points_gpu.cu:53-72, :14-50)
//
// The reference maps one thread to one OUTPUT element with the sample index
// fastest, so adjacent lanes read different feature rows (4-byte reads at a
// stride of C floats).  K5 here reads whole feature rows with the channel index
// on the lane (coalesced 256-B per wave-instruction), transposes a
// [C x 32-sample] tile through LDS and writes 128-B runs of the channel-major
// (M, C, nsample) output the reference's callers expect.  HBM-bound byte movers.
//
// The fused block kernels (fused_block.hip) never materialise these padded
// tensors; these entry points exist for API parity and as building blocks.
#include "common.hip.h"

#define GF_TILE_S 32  // samples per LDS tile
#define GF_TPB 256

__device__ __forceinline__ int batch_of_row(int B, int row, const int *idx_batch_cnt) {
    int bs = 0, cnt = idx_batch_cnt[0];  // ref group_features_gpu.cu:91-96
    for (int k = 1; k < B; ++k) {
        if (row < cnt) break;
        cnt += idx_batch_cnt[k];
        bs = k;
    }
    return bs;
}

__device__ __forceinline__ int feature_start(int bs, const int *features_batch_cnt) {
    int s = 0;  // ref :98-99
    for (int k = 0; k < bs; ++k) s += features_batch_cnt[k];
    return s;
}

// 1-D grid of M * ceil(nsample / 32) workgroups (no gridDim.y/z limit on the window
// count, unlike the reference's launches).  LDS tile [128][33] floats.
__global__ void __launch_bounds__(GF_TPB) k_group_features(int B, int M, int C, int nsample,
                                                           const float *features,
                                                           const int *features_batch_cnt,
                                                           const int *idx,
                                                           const int *idx_batch_cnt, float *out) {
    __shared__ float tile[128][GF_TILE_S + 1];
    __shared__ int s_idx[GF_TILE_S];
    __shared__ int s_start;
    const int ntiles = (nsample + GF_TILE_S - 1) / GF_TILE_S;
    const int m = blockIdx.x / ntiles;
    const int s0 = (blockIdx.x % ntiles) * GF_TILE_S;
    const int ns = min(GF_TILE_S, nsample - s0);
    if (threadIdx.x < GF_TILE_S)
        s_idx[threadIdx.x] = threadIdx.x < ns ? idx[(size_t)m * nsample + s0 + threadIdx.x] : -1;
    if (threadIdx.x == 0) s_start = feature_start(batch_of_row(B, m, idx_batch_cnt), features_batch_cnt);
    __syncthreads();
    const float *f = features + (size_t)s_start * C;
    const int lane = lane_id(), wv = threadIdx.x / MSSVT_WAVE;
    for (int c0 = 0; c0 < C; c0 += 128) {
        const int cn = min(128, C - c0);
        // load: wave wv takes samples wv, wv+4, ...; lanes sweep the channels of one row
        for (int s = wv; s < ns; s += GF_TPB / MSSVT_WAVE) {
            const int id = s_idx[s];
            for (int c = lane; c < cn; c += MSSVT_WAVE)
                tile[c][s] = id >= 0 ? f[(size_t)id * C + c0 + c] : 0.0f;
        }
        __syncthreads();
        // store: out[m, c0+c, s0+s], s fastest.  Slots with idx<0 are left untouched
        // (the caller pre-zeroed the output; ref :88).
        for (int e = threadIdx.x; e < cn * GF_TILE_S; e += GF_TPB) {
            const int c = e / GF_TILE_S, s = e % GF_TILE_S;
            if (s < ns && s_idx[s] >= 0)
                out[((size_t)m * C + c0 + c) * nsample + s0 + s] = tile[c][s];
        }
        __syncthreads();
    }
}

// Backward of K5: scatter-add.  One wave per (m, s) pair, lanes over channels so
// each atomic wave-instruction covers 256 contiguous bytes of one feature row
// (the shape that reaches the chip-wide float-atomic rate).
__global__ void __launch_bounds__(GF_TPB) k_group_features_grad(int B, int M, int C, int nsample,
                                                                const float *grad_out,
                                                                const int *idx,
                                                                const int *idx_batch_cnt,
                                                                const int *features_batch_cnt,
                                                                float *grad_features) {
    const long long pair = (long long)blockIdx.x * (GF_TPB / MSSVT_WAVE) + threadIdx.x / MSSVT_WAVE;
    if (pair >= (long long)M * nsample) return;
    const int m = (int)(pair / nsample), s = (int)(pair % nsample);
    const int id = idx[(size_t)m * nsample + s];
    if (id < 0) return;
    const int start = feature_start(batch_of_row(B, m, idx_batch_cnt), features_batch_cnt);
    float *g = grad_features + (size_t)(start + id) * C;
    const float *go = grad_out + (size_t)m * C * nsample + s;
    for (int c = lane_id(); c < C; c += MSSVT_WAVE) atomicAdd(g + c, go[(size_t)c * nsample]);
}

// K8: out[b,c,j] = points[b,c,idx[b,j]]
__global__ void k_gather_points(int b, int c, int n, int m, const float *points, const int *idx,
                                float *out) {
    const long long total = (long long)b * c * m;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e % m);
        const long long bc = e / m;
        const int bi = (int)(bc / c);
        out[e] = points[bc * n + idx[(size_t)bi * m + j]];
    }
}

__global__ void k_gather_points_grad(int b, int c, int n, int m, const float *grad_out,
                                     const int *idx, float *grad_points) {
    const long long total = (long long)b * c * m;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(e % m);
        const long long bc = e / m;
        const int bi = (int)(bc / c);
        atomicAdd(grad_points + bc * n + idx[(size_t)bi * m + j], grad_out[e]);
    }
}

// K10: out[b,c,p,s] = points[b,c,idx[b,p,s]]
__global__ void k_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                               const int *idx, float *out) {
    const long long per = (long long)npoints * nsample;
    const long long total = (long long)b * c * per;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long ps = e % per;
        const long long bc = e / per;
        const int bi = (int)(bc / c);
        out[e] = points[bc * n + idx[(size_t)bi * per + ps]];
    }
}

__global__ void k_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                    const float *grad_out, const int *idx, float *grad_points) {
    const long long per = (long long)npoints * nsample;
    const long long total = (long long)b * c * per;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long ps = e % per;
        const long long bc = e / per;
        const int bi = (int)(bc / c);
        atomicAdd(grad_points + bc * n + idx[(size_t)bi * per + ps], grad_out[e]);
    }
}

static inline int stride_grid(long long total, int tpb) {
    long long g = (total + tpb - 1) / tpb;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

extern "C" int mssvt_group_features(int B, int M, int C, int nsample, const float *features,
                                    const int *features_batch_cnt, const int *idx,
                                    const int *idx_batch_cnt, float *out, void *stream) {
    if (B <= 0 || M < 0 || C <= 0 || nsample <= 0) return MSSVT_E_BADARG;
    if (M == 0) return MSSVT_OK;
    if (!features || !features_batch_cnt || !idx || !idx_batch_cnt || !out) return MSSVT_E_BADARG;
    const long long nblk = (long long)M * divup(nsample, GF_TILE_S);
    if (nblk > 0x7FFFFFFFll) return MSSVT_E_TOOLARGE;
    k_group_features<<<(unsigned)nblk, GF_TPB, 0, (hipStream_t)stream>>>(B, M, C, nsample, features,
                                                               features_batch_cnt, idx,
                                                               idx_batch_cnt, out);
    return mssvt_launch_status();
}

extern "C" int mssvt_group_features_grad(int B, int M, int C, int N, int nsample,
                                         const float *grad_out, const int *idx,
                                         const int *idx_batch_cnt, const int *features_batch_cnt,
                                         float *grad_features, void *stream) {
    (void)N;
    if (B <= 0 || M < 0 || C <= 0 || nsample <= 0) return MSSVT_E_BADARG;
    if (M == 0) return MSSVT_OK;
    if (!grad_out || !idx || !idx_batch_cnt || !features_batch_cnt || !grad_features)
        return MSSVT_E_BADARG;
    const long long pairs = (long long)M * nsample;
    k_group_features_grad<<<divup(pairs, GF_TPB / MSSVT_WAVE), GF_TPB, 0, (hipStream_t)stream>>>(
        B, M, C, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features);
    return mssvt_launch_status();
}

extern "C" int mssvt_gather_points(int b, int c, int n, int npoints, const float *points,
                                   const int *idx, float *out, void *stream) {
    if (b < 0 || c <= 0 || n <= 0 || npoints <= 0) return MSSVT_E_BADARG;
    if (b == 0) return MSSVT_OK;
    if (!points || !idx || !out) return MSSVT_E_BADARG;
    const long long total = (long long)b * c * npoints;
    k_gather_points<<<stride_grid(total, 256), 256, 0, (hipStream_t)stream>>>(b, c, n, npoints,
                                                                             points, idx, out);
    return mssvt_launch_status();
}

extern "C" int mssvt_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                        const int *idx, float *grad_points, void *stream) {
    if (b < 0 || c <= 0 || n <= 0 || npoints <= 0) return MSSVT_E_BADARG;
    if (b == 0) return MSSVT_OK;
    if (!grad_out || !idx || !grad_points) return MSSVT_E_BADARG;
    const long long total = (long long)b * c * npoints;
    k_gather_points_grad<<<stride_grid(total, 256), 256, 0, (hipStream_t)stream>>>(
        b, c, n, npoints, grad_out, idx, grad_points);
    return mssvt_launch_status();
}

extern "C" int mssvt_group_points(int b, int c, int n, int npoints, int nsample,
                                  const float *points, const int *idx, float *out, void *stream) {
    if (b < 0 || c <= 0 || n <= 0 || npoints <= 0 || nsample <= 0) return MSSVT_E_BADARG;
    if (b == 0) return MSSVT_OK;
    if (!points || !idx || !out) return MSSVT_E_BADARG;
    const long long total = (long long)b * c * npoints * nsample;
    k_group_points<<<stride_grid(total, 256), 256, 0, (hipStream_t)stream>>>(b, c, n, npoints,
                                                                            nsample, points, idx, out);
    return mssvt_launch_status();
}

extern "C" int mssvt_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                       const float *grad_out, const int *idx, float *grad_points,
                                       void *stream) {
    if (b < 0 || c <= 0 || n <= 0 || npoints <= 0 || nsample <= 0) return MSSVT_E_BADARG;
    if (b == 0) return MSSVT_OK;
    if (!grad_out || !idx || !grad_points) return MSSVT_E_BADARG;
    const long long total = (long long)b * c * npoints * nsample;
    k_group_points_grad<<<stride_grid(total, 256), 256, 0, (hipStream_t)stream>>>(
        b, c, n, npoints, nsample, grad_out, idx, grad_points);
    return mssvt_launch_status();
}
