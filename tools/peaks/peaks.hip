// peaks.hip -- developer micro-benchmarks for the two ceilings the roofline numbers are quoted against
// (not part of libmssvt_hip): fp32 MFMA issue rate and HBM stream bandwidth on the box at hand.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

// `chains` independent v_mfma_f32_16x16x4_f32 accumulator chains per wave, `iters` rounds
// the inner loop of the row-tiled GEMMs of libmssvt_hip: per k-block of 16, CHAINS ds_read_b128 of padded
// weight rows (next block prefetched) + 4 * CHAINS MFMAs, B operand in registers
template <int CHAINS>
__global__ void __launch_bounds__(1024) k_mfma_lds(int iters, float *out) {
    extern __shared__ float4 lds4[];
    float *W = reinterpret_cast<float *>(lds4);
    constexpr int LS = 132, ROWS = 16 * CHAINS;
    for (int e = threadIdx.x; e < ROWS * LS; e += blockDim.x) W[e] = 1.0f + e * 1e-7f;
    __syncthreads();
    const int lane = threadIdx.x & 63, la = lane & 15, g = lane >> 4;
    f4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
    f4 x[8];
    for (int S = 0; S < 8; ++S) x[S] = f4{1.f + S, 2.f, 3.f + lane * 1e-6f, 4.f};
    const float *wbase = W + la * LS + 4 * g;
    for (int i = 0; i < iters; ++i) {
        float4 w[CHAINS], wn[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) w[c] = *reinterpret_cast<const float4 *>(wbase + c * 16 * LS);
#pragma unroll
        for (int S = 0; S < 8; ++S) {
            if (S + 1 < 8) {
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) wn[c] = *reinterpret_cast<const float4 *>(wbase + c * 16 * LS + 16 * (S + 1));
            }
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].x, x[S][0], acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].y, x[S][1], acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].z, x[S][2], acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c].w, x[S][3], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) w[c] = wn[c];
        }
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[0] = s;
}

template <int CHAINS>
__global__ void __launch_bounds__(1024) k_mfma_peak(int iters, float *out) {
    f4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[0] = s;  // keep the chains alive
}

__global__ void __launch_bounds__(256) k_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, long long n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_read(const float4 *__restrict__ src, float *out, long long n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < n4; i += stride) { const float4 v = src[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_write(float4 *__restrict__ dst, long long n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) dst[i] = float4{1.f, 2.f, 3.f, 4.f};
}

static float timed(hipStream_t st, int reps, void (*fn)(hipStream_t, void *), void *ctx) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    fn(st, ctx);
    hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) fn(st, ctx);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

struct MfmaCtx { int grid, iters, chains, threads, lds; float *out; };
static void run_mfma(hipStream_t st, void *p) {
    MfmaCtx *c = (MfmaCtx *)p;
    if (c->lds) {
        const size_t bytes = (size_t)16 * c->chains * 132 * 4;
        if (c->chains == 4) hipLaunchKernelGGL(k_mfma_lds<4>, dim3(c->grid), dim3(c->threads), bytes, st, c->iters, c->out);
        else hipLaunchKernelGGL(k_mfma_lds<8>, dim3(c->grid), dim3(c->threads), bytes, st, c->iters, c->out);
        return;
    }
    if (c->chains == 4) hipLaunchKernelGGL(k_mfma_peak<4>, dim3(c->grid), dim3(c->threads), 0, st, c->iters, c->out);
    else hipLaunchKernelGGL(k_mfma_peak<8>, dim3(c->grid), dim3(c->threads), 0, st, c->iters, c->out);
}
struct MemCtx { int grid, mode; float4 *a, *b; float *out; long long n4; };
static void run_mem(hipStream_t st, void *p) {
    MemCtx *c = (MemCtx *)p;
    if (c->mode == 0) hipLaunchKernelGGL(k_copy, dim3(c->grid), dim3(256), 0, st, c->a, c->b, c->n4);
    else if (c->mode == 1) hipLaunchKernelGGL(k_read, dim3(c->grid), dim3(256), 0, st, c->a, c->out, c->n4);
    else hipLaunchKernelGGL(k_write, dim3(c->grid), dim3(256), 0, st, c->b, c->n4);
}

extern "C" {
// TFLOP/s of dense fp32 MFMA with one workgroup of 4 * waves_per_simd waves per CU; lds = 1: with the weight
// fragment reads of the library's GEMM loops (iters = k-blocks of 32 MFMA steps), 0: MFMAs only
double peaks_mfma_f32(int cus, int waves_per_simd, int chains, int iters, int lds) {
    float *out; (void)hipMalloc(&out, 16);
    MfmaCtx c{cus, iters, chains, 256 * waves_per_simd, lds, out};
    const float ms = timed(0, 5, run_mfma, &c);
    (void)hipFree(out);
    const double per_iter = lds ? 32.0 * chains : (double)chains;
    const double flop = (double)c.grid * 4 * waves_per_simd * per_iter * iters * 2.0 * 16 * 16 * 4;
    return flop / (ms * 1e-3) / 1e12;
}
// GB/s moved (read + written bytes) by copy (mode 0), read-only (1), write-only (2) over `bytes` per array
double peaks_hbm(long long bytes, int mode, int grid) {
    float4 *a, *b; float *out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 16);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    MemCtx c{grid, mode, a, b, out, bytes / 16};
    const float ms = timed(0, 10, run_mem, &c);
    hipFree(a); hipFree(b); hipFree(out);
    const double moved = mode == 0 ? 2.0 * bytes : (double)bytes;
    return moved / (ms * 1e-3) / 1e9;
}
}
