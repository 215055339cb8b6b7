// fps_nn.hip -- K7 farthest point sampling and K9 three nearest neighbours.
//
// K7 replaces farthest_point_sampling_kernel<block_size>
// (ref: pointnet2/pointnet2_batch/src/sampling_gpu.cu:93-216, cuda_utils.h:10-14).
// The reference runs one CUDA block of bs = 2^floor(log2 N) threads per batch
// item and keeps its min-distance array in GLOBAL memory; for the window sizes on
// the path (N = 45 / 343 / 27 ...) that is 32..256 threads doing 31 rounds of a
// strided global scan plus a __syncthreads tree.  Here ONE WAVEFRONT owns a batch
// item: points and running min-distances live in LDS as float4, each lane plays
// the reference threads tid = lane, lane+64, ..., and the reference's shared-memory
// tree is replayed level by level (LDS for strides >= 64, __shfl_down below) so
// that every arg-max TIE resolves to the same index the CUDA block would pick
// (ties are the norm: the inputs are small integer offsets plus zero padding).
//
// K9 replaces three_nn_kernel_fast (ref: .../interpolate_gpu.cu:16-59): one lane
// per unknown point, the known points are read with wave-uniform (broadcast)
// addresses.  The squared distance is the fma chain nvcc's default -fmad=true
// emits for a*a+b*b+c*c, written explicitly so that it matches the oracle.
#include "common.hip.h"
#include <math.h>

#define FPS_WPB 4  // waves (= batch items) per workgroup

__device__ __forceinline__ float sqdist(float x1, float y1, float z1, float x2, float y2, float z2) {
    const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// opt_n_threads(): ref cuda_utils.h:10-14 (host-side double arithmetic, kept as is)
static inline int opt_n_threads(int work_size) {
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

// LDS layout per wave: float4 pts[n] | float bv[bs] | int bi[bs]   (bv/bi only if bs > 64)
template <bool POINTS_IN_LDS>
__global__ void __launch_bounds__(FPS_WPB *MSSVT_WAVE)
    k_fps(int b, int n, int m, int bs, const float *__restrict__ dataset, float *__restrict__ temp,
          int *__restrict__ idxs, int lds_floats_per_wave) {
    extern __shared__ float4 lds4[];
    const int wv = threadIdx.x / MSSVT_WAVE, lane = lane_id();
    const int bi = blockIdx.x * FPS_WPB + wv;
    if (bi >= b) return;  // wave-uniform; no workgroup barriers below
    float *wbase = reinterpret_cast<float *>(lds4) + (size_t)wv * lds_floats_per_wave;
    float4 *pts = reinterpret_cast<float4 *>(wbase);
    float *bv = wbase + (POINTS_IN_LDS ? 4 * n : 0);
    int *bidx = reinterpret_cast<int *>(bv + bs);
    const float *d = dataset + (size_t)bi * n * 3;
    float *tmp = temp + (size_t)bi * n;
    int *out = idxs + (size_t)bi * m;

    if (POINTS_IN_LDS) {
        for (int k = lane; k < n; k += MSSVT_WAVE)
            pts[k] = make_float4(d[k * 3 + 0], d[k * 3 + 1], d[k * 3 + 2], tmp[k]);
        wave_lds_sync();
    }
    int old = 0;
    if (lane == 0) out[0] = 0;
    const int top = bs < MSSVT_WAVE ? bs : MSSVT_WAVE;  // width of the shuffle part of the tree
    for (int j = 1; j < m; ++j) {
        float x1, y1, z1;
        if (POINTS_IN_LDS) {
            const float4 p = pts[old];
            x1 = p.x; y1 = p.y; z1 = p.z;
        } else {
            x1 = d[old * 3 + 0]; y1 = d[old * 3 + 1]; z1 = d[old * 3 + 2];
        }
        float best = -1.0f;
        int besti = 0;
        for (int vt = lane; vt < bs; vt += MSSVT_WAVE) {  // reference thread `vt`
            best = -1.0f;
            besti = 0;
            for (int k = vt; k < n; k += bs) {  // ref :131-145
                float d2;
                if (POINTS_IN_LDS) {
                    float4 p = pts[k];
                    d2 = fminf(sqdist(x1, y1, z1, p.x, p.y, p.z), p.w);
                    pts[k].w = d2;
                } else {
                    d2 = fminf(sqdist(x1, y1, z1, d[k * 3 + 0], d[k * 3 + 1], d[k * 3 + 2]), tmp[k]);
                    tmp[k] = d2;
                }
                besti = d2 > best ? k : besti;
                best = d2 > best ? d2 : best;
            }
            if (bs > MSSVT_WAVE) {
                bv[vt] = best;
                bidx[vt] = besti;
            }
        }
        // tree levels with stride >= 64: pairs (t, t+s) live in LDS (ref :149-170)
        for (int s = bs >> 1; s >= MSSVT_WAVE; s >>= 1) {
            wave_lds_sync();
            for (int t = lane; t < s; t += MSSVT_WAVE) {
                const float v1 = bv[t], v2 = bv[t + s];
                const int i1 = bidx[t], i2 = bidx[t + s];
                bv[t] = fmaxf(v1, v2);
                bidx[t] = v2 > v1 ? i2 : i1;  // ref __update :93-98
            }
        }
        if (bs > MSSVT_WAVE) {
            wave_lds_sync();
            best = bv[lane];
            besti = bidx[lane];
        }
        // remaining levels (stride < 64) across lanes (ref :171-208)
        for (int s = top >> 1; s >= 1; s >>= 1) {
            const float v2 = __shfl_down(best, s);
            const int i2 = __shfl_down(besti, s);
            besti = v2 > best ? i2 : besti;
            best = fmaxf(best, v2);
        }
        old = __builtin_amdgcn_readfirstlane(besti);
        if (lane == 0) out[j] = old;
        if (POINTS_IN_LDS) wave_lds_sync();  // pts[].w updates vs. next round's reads
    }
    if (POINTS_IN_LDS) {  // keep `temp` as the reference leaves it (scratch, but observable)
        for (int k = lane; k < n; k += MSSVT_WAVE) tmp[k] = pts[k].w;
    }
}

extern "C" int mssvt_farthest_point_sampling(int b, int n, int m, const float *dataset,
                                             float *temp, int *idxs, void *stream) {
    if (b < 0 || n <= 0 || m < 0) return MSSVT_E_BADARG;
    if (b == 0 || m == 0) return MSSVT_OK;  // ref :108 `if (m <= 0) return`
    if (!dataset || !temp || !idxs) return MSSVT_E_BADARG;
    const int bs = opt_n_threads(n);
    const int tree_floats = bs > MSSVT_WAVE ? 2 * bs : 0;
    int per_wave = 4 * n + tree_floats;
    per_wave = (per_wave + 3) & ~3;  // keep each wave's float4 region 16-B aligned
    const bool in_lds = (size_t)per_wave * sizeof(float) * FPS_WPB <= 96 * 1024;
    const int grid = divup(b, FPS_WPB);
    if (in_lds) {
        k_fps<true><<<grid, FPS_WPB * MSSVT_WAVE, (size_t)per_wave * sizeof(float) * FPS_WPB,
                      (hipStream_t)stream>>>(b, n, m, bs, dataset, temp, idxs, per_wave);
    } else {
        per_wave = (tree_floats + 3) & ~3;
        k_fps<false><<<grid, FPS_WPB * MSSVT_WAVE, (size_t)per_wave * sizeof(float) * FPS_WPB,
                       (hipStream_t)stream>>>(b, n, m, bs, dataset, temp, idxs, per_wave);
    }
    return mssvt_launch_status();
}

// ---------------------------------------------------------------------------
// K9
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_three_nn(int b, int n, int m, int chunks_per_item,
                                                  const float *__restrict__ unknown,
                                                  const float *__restrict__ known,
                                                  float *__restrict__ dist2, int *__restrict__ idx) {
    const long long item = (long long)blockIdx.x * 4 + threadIdx.x / MSSVT_WAVE;  // wave work item
    if (item >= (long long)b * chunks_per_item) return;
    const int bi = (int)(item / chunks_per_item);
    const int pt = (int)(item % chunks_per_item) * MSSVT_WAVE + lane_id();
    if (pt >= n) return;
    const float *u = unknown + ((size_t)bi * n + pt) * 3;
    const float *kn = known + (size_t)bi * m * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    double best1 = 1e40, best2 = 1e40, best3 = 1e40;  // ref :37
    int besti1 = 0, besti2 = 0, besti3 = 0;
    for (int k = 0; k < m; ++k) {
        const float dx = ux - kn[k * 3 + 0], dy = uy - kn[k * 3 + 1], dz = uz - kn[k * 3 + 2];
        const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        if (d < best1) {
            best3 = best2; besti3 = besti2;
            best2 = best1; besti2 = besti1;
            best1 = d; besti1 = k;
        } else if (d < best2) {
            best3 = best2; besti3 = besti2;
            best2 = d; besti2 = k;
        } else if (d < best3) {
            best3 = d; besti3 = k;
        }
    }
    float *o = dist2 + ((size_t)bi * n + pt) * 3;
    int *oi = idx + ((size_t)bi * n + pt) * 3;
    o[0] = (float)best1; o[1] = (float)best2; o[2] = (float)best3;
    oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
}

extern "C" int mssvt_three_nn(int b, int n, int m, const float *unknown, const float *known,
                              float *dist2, int *idx, void *stream) {
    if (b < 0 || n <= 0 || m < 0) return MSSVT_E_BADARG;
    if (b == 0) return MSSVT_OK;
    if (!unknown || !known || !dist2 || !idx) return MSSVT_E_BADARG;
    const int chunks = divup(n, MSSVT_WAVE);
    const long long items = (long long)b * chunks;
    k_three_nn<<<divup(items, 4), 256, 0, (hipStream_t)stream>>>(b, n, m, chunks, unknown, known,
                                                                 dist2, idx);
    return mssvt_launch_status();
}
