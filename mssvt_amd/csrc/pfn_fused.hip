// pfn_fused.hip -- the two PFN layers of DynamicVFE (eval mode) as two launches, for the default configuration
// (5 point features + cluster offset + voxel-centre offset = 11 inputs, NUM_FILTERS [64, 128]; ref
// pcdet/models/backbones_3d/vfe/dynamic_vfe.py:96-131, PFNLayerV2 :14-52 -- SURVEY.md section 8 f1):
//
//     f    = [x, y, z, i, e,  xyz - mean_xyz[voxel],  xyz - centre(voxel)]                               (P, 11)
//     x1   = relu(bn1(W1 f + b1))                                                                        (P, 64)
//     m1   = scatter_max(x1, voxel)                                                                      (N, 64)
//     x2   = relu(bn2(W2 [x1 ; m1[voxel]] + b2))                                                         (P, 128)
//     out  = scatter_max(x2, voxel)                                                                      (N, 128)
//
// The module path runs this as ~25 framework launches and five passes over (P, 64..128) tensors per layer (gather, cat,
// library GEMM, batch-norm transform, clamp, scatter-max: 0.6 ms at 160k points -- as much as the whole backbone).  Here:
//   k_pfn1    16 lanes per point (4 output channels each): builds the 11 inputs on the fly from the point row, the voxel's
//             mean and its integer coordinate (the reference's expressions, multiply and add rounded separately), 44 FMAs per
//             lane, BatchNorm (eval: running statistics) + ReLU, writes x1;
//   k_pfn_rowmax  the per-voxel max of x1 / x2 (order-preserving integer atomics: order independent, deterministic; one
//             wavefront per point so that an atomic instruction covers 256 contiguous bytes of one voxel row);
//   k_pfn2_h  the 128 -> 128 layer on split-fp16 matrix operands (the arithmetic and tile structure of csrc/linear_rows_h.hip:
//             the weight matrix as hi / lo MFMA fragments in LDS, 16-point tiles, rows normalised by a power of two): a tile's
//             rows are gathered as [x1[p] ; m1[voxel[p]]], the epilogue is BatchNorm + ReLU, x2 leaves as 16-byte stores.
// Five launches (fill, k_pfn1, max, k_pfn2_h, max) instead of ~25.
// Points outside the grid (voxel < 0) are skipped.  Parity: tests/test_vfe_gpu.py (the reference-run goldens and the numpy
// restatement at full size, 1e-4 of scale: products re-associated, BatchNorm applied in torch's operation order).
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
#define PF_SCALE 2048.0f
#define PF_INV (1.0f / 2048.0f)
#define PF_MFMA(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16((av), (bv), acc, 0, 0, 0)

__device__ __forceinline__ void pf_atomic_max(float *addr, float v) {  // (as csrc/vfe.hip: order-preserving integer views)
    if (v >= 0.0f)
        atomicMax(reinterpret_cast<int *>(addr), __builtin_bit_cast(int, v));
    else
        atomicMin(reinterpret_cast<unsigned int *>(addr), __builtin_bit_cast(unsigned int, v));
}

struct Pfn1Args {
    const float *points;  // (P, stride) rows [b, x, y, z, f4, f5]
    int stride;
    long long P;
    const int *voxel;    // (P) voxel of the point, -1 outside the grid
    const float *mean3;  // (N, 3)
    const int *coords;   // (N, 4) [b, z, y, x]
    float vs[3], off[3];  // voxel size, voxel_size / 2 + range_min
    const float *W, *b, *bn_w, *bn_b, *bn_mean, *bn_var;  // W (64, 11)
    float eps;
    float *x1, *m1;  // (P, 64), (N, 64) pre-filled with -inf
};

__global__ void __launch_bounds__(256) k_pfn1(Pfn1Args a) {
    __shared__ float Wl[64 * 12], bl[64], sl[64], tl[64];
    for (int e = threadIdx.x; e < 64 * 11; e += 256) Wl[(e / 11) * 12 + e % 11] = a.W[e];
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        bl[c] = a.b[c];
        // torch's eval batch norm: (z - mean) * invstd * weight + bias, invstd = 1 / sqrt(var + eps)
        sl[c] = 1.0f / sqrtf(a.bn_var[c] + a.eps);
        tl[c] = a.bn_mean[c];
    }
    __syncthreads();
    const int q = threadIdx.x & 15;  // channels [4 q, 4 q + 4) of the point
    const long long p = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= a.P) return;
    const int v = a.voxel[p];
    if (v < 0) return;
    const float *pr = a.points + p * a.stride;
    const float x = pr[1], y = pr[2], z = pr[3];
    const int4 c4 = reinterpret_cast<const int4 *>(a.coords)[v];
    float f[11];
    f[0] = x; f[1] = y; f[2] = z; f[3] = pr[4]; f[4] = pr[5];
    f[5] = x - a.mean3[(size_t)v * 3 + 0]; f[6] = y - a.mean3[(size_t)v * 3 + 1]; f[7] = z - a.mean3[(size_t)v * 3 + 2];
    f[8] = x - __fadd_rn(__fmul_rn((float)c4.w, a.vs[0]), a.off[0]);  // ref :107-109: coord * voxel_size + offset
    f[9] = y - __fadd_rn(__fmul_rn((float)c4.z, a.vs[1]), a.off[1]);
    f[10] = z - __fadd_rn(__fmul_rn((float)c4.y, a.vs[2]), a.off[2]);
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * q + i;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) s = __builtin_fmaf(f[k], Wl[c * 12 + k], s);
        s += bl[c];
        s = (s - tl[c]) * sl[c] * a.bn_w[c] + a.bn_b[c];
        o[i] = fmaxf(s, 0.f);
    }
    *reinterpret_cast<float4 *>(a.x1 + p * 64 + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ void pf_split8(const float4 v0, const float4 v1, float s, h16x8 &hi, h16x8 &lo) {
    const float x[8] = {v0.x * s, v0.y * s, v0.z * s, v0.w * s, v1.x * s, v1.y * s, v1.z * s, v1.w * s};
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const fp16x2 a = __builtin_amdgcn_cvt_pkrtz(x[i], x[i + 1]);
        const fp16x2 c = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)a[0], -PF_SCALE, x[i] * PF_SCALE),
                                                    __builtin_fmaf((float)a[1], -PF_SCALE, x[i + 1] * PF_SCALE));
        hi[i] = (_Float16)a[0]; hi[i + 1] = (_Float16)a[1];
        lo[i] = (_Float16)c[0]; lo[i + 1] = (_Float16)c[1];
    }
}

struct Pfn2Args {
    long long P;
    const int *voxel;
    const float *x1, *m1;  // (P, 64), (N, 64)
    const float *W, *b, *bn_w, *bn_b, *bn_mean, *bn_var;  // W (128, 128): columns [0, 64) <-> x1, [64, 128) <-> m1[voxel]
    float eps;
    float *out;  // x2 (P, 128)
};

#define PF2_WAVES 16
__global__ void __launch_bounds__(PF2_WAVES *MSSVT_WAVE, 1) k_pfn2_h(Pfn2Args a) {
    constexpr int K = 128, N = 128, KS = K / 32, NT = N / 16, IMG = KS * NT * 64;
    extern __shared__ float4 lds4[];
    h16x8 *Bh = reinterpret_cast<h16x8 *>(lds4), *Bl = Bh + IMG;
    __shared__ float wmax_l[PF2_WAVES];
    __shared__ float4 ep_l[3][N / 4];  // per channel: bias - mean | invstd * weight | bn bias
    constexpr int FR = IMG / (PF2_WAVES * MSSVT_WAVE);  // 2 fragments per thread
    float4 wv0[FR], wv1[FR];
    float wmx = 0.f;
#pragma unroll
    for (int i = 0; i < FR; ++i) {
        const int f = threadIdx.x + i * PF2_WAVES * MSSVT_WAVE;
        const int ln = f & 63, t = (f >> 6) % NT, P = (f >> 6) / NT;
        const int n = 16 * t + (ln & 15), k0 = 32 * P + 8 * (ln >> 4);
        wv0[i] = *reinterpret_cast<const float4 *>(a.W + (size_t)n * K + k0);
        wv1[i] = *reinterpret_cast<const float4 *>(a.W + (size_t)n * K + k0 + 4);
        wmx = fmaxf(wmx, fmaxf(fmaxf(fmaxf(fabsf(wv0[i].x), fabsf(wv0[i].y)), fmaxf(fabsf(wv0[i].z), fabsf(wv0[i].w))),
                               fmaxf(fmaxf(fabsf(wv1[i].x), fabsf(wv1[i].y)), fmaxf(fabsf(wv1[i].z), fabsf(wv1[i].w)))));
    }
    if (threadIdx.x < N) {
        const int c = threadIdx.x;
        reinterpret_cast<float *>(ep_l[0])[c] = a.b[c];
        reinterpret_cast<float *>(ep_l[1])[c] = a.bn_mean[c];
        reinterpret_cast<float *>(ep_l[2])[c] = 1.0f / sqrtf(a.bn_var[c] + a.eps);
    }
    wmx = wave_max(wmx);
    if (lane_id() == 0) wmax_l[threadIdx.x / MSSVT_WAVE] = wmx;
    __syncthreads();
    wmx = 0.f;
#pragma unroll
    for (int i = 0; i < PF2_WAVES; ++i) wmx = fmaxf(wmx, wmax_l[i]);
    const int web = __builtin_bit_cast(int, wmx) & 0x7F800000;
    const bool wnorm = web != 0 && web < 0x7F000000;
    const float w_in = wnorm ? __builtin_bit_cast(float, 0x7F000000 - web) : 1.0f;
    const float w_un = wnorm ? __builtin_bit_cast(float, web) : 1.0f;
#pragma unroll
    for (int i = 0; i < FR; ++i) {
        const int f = threadIdx.x + i * PF2_WAVES * MSSVT_WAVE;
        h16x8 h, l;
        pf_split8(wv0[i], wv1[i], w_in, h, l);
        Bh[f] = h;
        Bl[f] = l;
    }
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    __syncthreads();
    const long long tiles = (a.P + 15) / 16, step = (long long)gridDim.x * PF2_WAVES;
    for (long long tile = (long long)blockIdx.x * PF2_WAVES + wv; tile < tiles; tile += step) {
        const long long p = min(tile * 16 + la, a.P - 1);
        const int v = tile * 16 + la < a.P ? a.voxel[p] : -1;
        const int vs = max(v, 0);
        // lane (p = la, g) reads its row's k slots 32 P + 8 g ..: P = 0, 1 from x1[p], P = 2, 3 from m1[voxel]
        float4 xr[KS][2];
#pragma unroll
        for (int P = 0; P < KS; ++P) {
            const float *src = P < 2 ? a.x1 + p * 64 + 32 * P + 8 * g : a.m1 + (size_t)vs * 64 + 32 * (P - 2) + 8 * g;
            xr[P][0] = *reinterpret_cast<const float4 *>(src);
            xr[P][1] = *reinterpret_cast<const float4 *>(src + 4);
        }
        float mx = 0.f;
#pragma unroll
        for (int P = 0; P < KS; ++P)
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(xr[P][0].x), fabsf(xr[P][0].y)), fmaxf(fabsf(xr[P][0].z), fabsf(xr[P][0].w))),
                                 fmaxf(fmaxf(fabsf(xr[P][1].x), fabsf(xr[P][1].y)), fmaxf(fabsf(xr[P][1].z), fabsf(xr[P][1].w)))));
        mx = fmaxf(mx, lane_xor16(mx));
        mx = fmaxf(mx, lane_xor32(mx));
        const int eb = __builtin_bit_cast(int, mx) & 0x7F800000;
        const bool norm = eb != 0 && eb < 0x7F000000;
        const float s_in = norm ? __builtin_bit_cast(float, 0x7F000000 - eb) : 1.0f;
        const float un = (norm ? __builtin_bit_cast(float, eb) : 1.0f) * w_un;
        h16x8 ah[KS], al[KS];
#pragma unroll
        for (int P = 0; P < KS; ++P) pf_split8(xr[P][0], xr[P][1], s_in, ah[P], al[P]);
        float *orow = a.out + (size_t)p * N + 4 * g;  // x2[p]: reduced per voxel by k_pfn_rowmax (256 contiguous bytes per atomic instruction)
#pragma unroll 2
        for (int t = 0; t < NT; ++t) {
            f32x4 mm = f32x4{0.f, 0.f, 0.f, 0.f}, cr = mm;
#pragma unroll
            for (int P = 0; P < KS; ++P) {
                const h16x8 bh = Bh[(P * NT + t) * 64 + lane], bl = Bl[(P * NT + t) * 64 + lane];
                PF_MFMA(mm, bh, ah[P]);
                PF_MFMA(cr, bh, al[P]);
                PF_MFMA(cr, bl, ah[P]);
            }
            // lane (p = la, g) holds x2[p][16 t + 4 g + i]
            const float4 b4 = ep_l[0][4 * t + g], m4 = ep_l[1][4 * t + g], s4 = ep_l[2][4 * t + g];
            const float4 w4 = *reinterpret_cast<const float4 *>(a.bn_w + 16 * t + 4 * g),
                         c4 = *reinterpret_cast<const float4 *>(a.bn_b + 16 * t + 4 * g);
            float r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(cr[i], PF_INV, mm[i]) * un;
            const float o0 = fmaxf(((r[0] + b4.x) - m4.x) * s4.x * w4.x + c4.x, 0.f), o1 = fmaxf(((r[1] + b4.y) - m4.y) * s4.y * w4.y + c4.y, 0.f),
                        o2 = fmaxf(((r[2] + b4.z) - m4.z) * s4.z * w4.z + c4.z, 0.f), o3 = fmaxf(((r[3] + b4.w) - m4.w) * s4.w * w4.w + c4.w, 0.f);
            if (v >= 0) *reinterpret_cast<float4 *>(orow + 16 * t) = make_float4(o0, o1, o2, o3);
        }
    }
}

// scatter-max of rows into their voxels: one wavefront per point, lanes over channels -- every atomic instruction covers 256
// contiguous bytes of ONE voxel row (the memory-side atomic units run those at full rate; the accumulator layout of the product
// would scatter a wave's 64 dwords over 16 rows: 235 us for this reduction instead of 45)
template <int F>
__global__ void __launch_bounds__(256) k_pfn_rowmax(const float *x, long long P, const int *voxel, float *out) {
    const long long p = (long long)blockIdx.x * 4 + threadIdx.x / MSSVT_WAVE;
    if (p >= P) return;
    const int v = voxel[p];
    if (v < 0) return;
#pragma unroll
    for (int c = lane_id(); c < F; c += MSSVT_WAVE) pf_atomic_max(out + (size_t)v * F + c, x[p * F + c]);
}

__global__ void __launch_bounds__(256) k_pfn_fill2(float *a, long long na, float *b, long long nb) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < na) a[i] = -INFINITY;
    if (i < nb) b[i] = -INFINITY;
}

// DynamicVFE's two PFN layers in eval mode (ref dynamic_vfe.py:96-131), default configuration: 5 point features, cluster
// and voxel-centre offsets, NUM_FILTERS [64, 128].  points (P, stride >= 6) f32 rows [b, x, y, z, f4, f5]; point_voxel (P)
// int32 (-1: outside the grid); mean3 (N, 3) f32 = scatter_mean of xyz (mssvt_voxel_mean_xyz); voxel_coords (N, 4) int32
// [b, z, y, x]; host_voxel_size3 / host_offset3: HOST float[3] (offset = voxel_size / 2 + range_min); layer parameters as the
// state dict holds them (pfn.{0,1}.0.weight / .bias, pfn.{0,1}.1.weight / .bias / .running_mean / .running_var);
// x1_scratch (P, 64), m1_scratch (max(N, 1), 64), x2_scratch (P, 128): caller-owned; out (N, 128): the voxel features.
extern "C" int mssvt_pfn_fused_64_128(const float *points, int point_stride, long long num_points, const int *point_voxel,
                                      int num_voxels, const float *mean3, const int *voxel_coords, const float *host_voxel_size3,
                                      const float *host_offset3, const float *W1, const float *b1, const float *bn1_w,
                                      const float *bn1_b, const float *bn1_mean, const float *bn1_var, float bn1_eps, const float *W2,
                                      const float *b2, const float *bn2_w, const float *bn2_b, const float *bn2_mean,
                                      const float *bn2_var, float bn2_eps, float *x1_scratch, float *m1_scratch, float *x2_scratch,
                                      float *out, void *stream_) {
    if (num_points < 0 || num_voxels < 0 || point_stride < 6 || (num_points > 0 && (!points || !point_voxel)) || !mean3 || !voxel_coords ||
        !host_voxel_size3 || !host_offset3 || !W1 || !b1 || !bn1_w || !bn1_b || !bn1_mean || !bn1_var || !W2 || !b2 || !bn2_w || !bn2_b ||
        !bn2_mean || !bn2_var || !x1_scratch || !m1_scratch || !x2_scratch || !out)
        return MSSVT_E_BADARG;
    if (num_voxels == 0) return MSSVT_OK;
    hipStream_t stream = (hipStream_t)stream_;
    const long long n1 = (long long)num_voxels * 64, n2 = (long long)num_voxels * 128;
    k_pfn_fill2<<<divup(n2, 256), 256, 0, stream>>>(m1_scratch, n1, out, n2);
    if (num_points == 0) return mssvt_launch_status();
    Pfn1Args a1;
    a1.points = points; a1.stride = point_stride; a1.P = num_points; a1.voxel = point_voxel; a1.mean3 = mean3; a1.coords = voxel_coords;
    for (int k = 0; k < 3; ++k) { a1.vs[k] = host_voxel_size3[k]; a1.off[k] = host_offset3[k]; }
    a1.W = W1; a1.b = b1; a1.bn_w = bn1_w; a1.bn_b = bn1_b; a1.bn_mean = bn1_mean; a1.bn_var = bn1_var; a1.eps = bn1_eps;
    a1.x1 = x1_scratch; a1.m1 = m1_scratch;
    k_pfn1<<<divup(num_points, 16), 256, 0, stream>>>(a1);
    k_pfn_rowmax<64><<<divup(num_points, 4), 256, 0, stream>>>(x1_scratch, num_points, point_voxel, m1_scratch);
    Pfn2Args a2;
    a2.P = num_points; a2.voxel = point_voxel; a2.x1 = x1_scratch; a2.m1 = m1_scratch;
    a2.W = W2; a2.b = b2; a2.bn_w = bn2_w; a2.bn_b = bn2_b; a2.bn_mean = bn2_mean; a2.bn_var = bn2_var; a2.eps = bn2_eps; a2.out = x2_scratch;
    const size_t lds = (size_t)128 * 128 * 4;
    {   // (per device and cheap next to the launches: set on every call, as the other launchers do)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_pfn2_h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int grid = min((long long)cus, (long long)divup(num_points, 16 * PF2_WAVES));
    k_pfn2_h<<<grid, PF2_WAVES * MSSVT_WAVE, lds, stream>>>(a2);
    k_pfn_rowmax<128><<<divup(num_points, 4), 256, 0, stream>>>(x2_scratch, num_points, point_voxel, out);
    return mssvt_launch_status();
}
