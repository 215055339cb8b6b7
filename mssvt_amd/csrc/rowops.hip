// rowops.hip -- small per-row helpers of the fused path that would otherwise be framework launches.
//
//   mssvt_batch_counts : rows per sample of a (N,4) [b,z,y,x] index tensor.  The reference loops over
//                        the samples on the host with .item() (ref: mssvt_utils.py:35-37,
//                        mssvt_backbone.py:124-130); a framework bincount costs two reductions
//                        (min / max), a histogram and a host synchronisation.
//   mssvt_layer_norm   : norm1 of the first block (ref: mssvt_backbone.py:241; every later norm1 is
//                        emitted by the previous block's fused FFN epilogue, csrc/ffn.hip).
#include "common.hip.h"

#define BC_ROWS 16   // rows per thread
#define BC_LDS 1024  // samples counted in LDS per workgroup
__global__ void __launch_bounds__(256) k_batch_counts(const int *indices, int n, int batch_size, int *counts) {
    // one global atomic per (workgroup, sample present in it): single-address atomics cost ~11 ns each
    // chip-wide -- one per wavefront would be 1 161 of them for a 74k-voxel scene, one per row of a
    // workgroup that straddles a sample boundary 4 096
    __shared__ int cnt[BC_LDS];
    __shared__ int first_cnt;
    for (int e = threadIdx.x; e < BC_LDS; e += 256) cnt[e] = 0;
    if (threadIdx.x == 0) first_cnt = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * 256 * BC_ROWS;
    // samples are contiguous: nearly every workgroup sees one sample only -> count it in registers
    const int b_first = base < n ? indices[4 * base] : -1;
    int local = 0;
#pragma unroll 4
    for (int r = 0; r < BC_ROWS; ++r) {
        const long long i = base + (long long)r * 256 + threadIdx.x;
        if (i >= n) break;
        const int b = indices[4 * i];
        if (b < 0 || b >= batch_size) continue;
        if (b == b_first) ++local;
        else if (b < BC_LDS) atomicAdd(&cnt[b], 1);
        else atomicAdd(counts + b, 1);
    }
    local = wave_sum_i(local);
    if (lane_id() == 0 && local) atomicAdd(&first_cnt, local);
    __syncthreads();
    if (threadIdx.x == 0 && first_cnt && b_first >= 0 && b_first < batch_size) atomicAdd(counts + b_first, first_cnt);
    for (int e = threadIdx.x; e < BC_LDS && e < batch_size; e += 256)
        if (cnt[e]) atomicAdd(counts + e, cnt[e]);
}

extern "C" int mssvt_batch_counts(const int *indices, int num_rows, int batch_size, int *counts, void *stream_) {
    if (!counts || (!indices && num_rows > 0) || num_rows < 0 || batch_size <= 0) return MSSVT_E_BADARG;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)batch_size * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    return mssvt_batch_counts_launch(indices, num_rows, batch_size, counts, stream);
}

int mssvt_batch_counts_launch(const int *indices, int num_rows, int batch_size, int *counts, hipStream_t stream) {
    if (num_rows > 0)
        k_batch_counts<<<divup(num_rows, 256 * BC_ROWS), 256, 0, stream>>>(indices, num_rows, batch_size, counts);
    return mssvt_launch_status();
}

// LPR lanes (float4 each) per row, 64 / LPR rows per wavefront instruction
template <int LPR>
__global__ void __launch_bounds__(256) k_layer_norm(const float *x, int n, const float *w, const float *b, float eps,
                                                    float *y) {
    constexpr int C = LPR * 4, RPW = MSSVT_WAVE / LPR;
    const int lane = lane_id();
    const size_t row = ((size_t)blockIdx.x * 4 + threadIdx.x / MSSVT_WAVE) * RPW + lane / LPR;
    const int col = (lane % LPR) * 4;
    const bool live = row < (size_t)n;
    const float4 v = live ? *reinterpret_cast<const float4 *>(x + row * C + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float m = s * (1.0f / C);
    const float dx = v.x - m, dy = v.y - m, dz = v.z - m, dw = v.w - m;
    float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rs = rsqrtf(q * (1.0f / C) + eps);
    const float4 g4 = *reinterpret_cast<const float4 *>(w + col);
    const float4 b4 = *reinterpret_cast<const float4 *>(b + col);
    if (live)
        *reinterpret_cast<float4 *>(y + row * C + col) =
            make_float4(dx * rs * g4.x + b4.x, dy * rs * g4.y + b4.y, dz * rs * g4.z + b4.z, dw * rs * g4.w + b4.w);
}

extern "C" int mssvt_layer_norm(const float *x, int num_rows, int C, const float *weight, const float *bias,
                                float eps, float *y, void *stream_) {
    if (!x || !weight || !bias || !y || num_rows < 0 || C <= 0) return MSSVT_E_BADARG;
    if (num_rows == 0) return MSSVT_OK;
    hipStream_t stream = (hipStream_t)stream_;
#define LN_CASE(lpr)                                                                                         \
    if (C == 4 * lpr) {                                                                                      \
        const int rows_per_block = 4 * (MSSVT_WAVE / lpr);                                                   \
        k_layer_norm<lpr><<<divup(num_rows, rows_per_block), 256, 0, stream>>>(x, num_rows, weight, bias, eps, y); \
        return mssvt_launch_status();                                                                        \
    }
    LN_CASE(4)
    LN_CASE(8)
    LN_CASE(16)
    LN_CASE(32)
    LN_CASE(64)
#undef LN_CASE
    return MSSVT_E_TOOLARGE;  // C not in {16, 32, 64, 128, 256}: the caller uses the framework's LayerNorm
}
