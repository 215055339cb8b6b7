// linear_rows.hip -- Y = X B^T + bias over compact rows on the fp32 matrix cores, for the SMALL weight matrices of a Block
// (64 x 64: to_qs and projs of a head group; training path): the forward of the nn.Linear
// (ref mssvt_utils.py:80-83) and, with the weight read transposed, its input gradient dX = dY W.
//
//     Y[m][n] = s * act(bias[n] + sum_k X[m][k] * B[n][k])   B = W (N x K row-major), or B[n][k] = W[k][n] (transpose_w)
// (s = out_scale: the attention scale of the query projection rides in the epilogue, forward and backward)
//
// The library GEMM the framework calls is fine on the Blocks' large shapes (128 <-> 256: 47 - 85 TFLOP/s; a kernel of
// this form was measured at 0.4 - 0.7 of that and is not used there) but not on 64 x 64: 13 us for 33k rows, and 154 us
// for 132k rows (7 TFLOP/s: an unlucky tile choice at four scenes per step) against 8.4 / 34 us here, plus ~26 us of host
// time per call for its algorithm lookup against ~5.  One workgroup of 8 waves per CU stages B once ([n][k], rows padded
// by 4 floats) and streams 16-row tiles: a lane loads 16 bytes of its row per 16 columns of K (the next tile's loads in
// flight under the current tile's products), the B operand is one 16-byte LDS read per four v_mfma_f32_16x16x4_f32 -- A and
// B agree on which four k a lane pair holds, so no operand is rearranged.
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LR_WAVES 8
#define LR_MFMA(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av), (bv), acc, 0, 0, 0)

template <int K, int NT>  // K columns of X, NT = N / 16 column tiles of Y
__global__ void __launch_bounds__(LR_WAVES *MSSVT_WAVE, 1) k_linear_rows(int M, const float *X, int ldx, const float *W, int transpose_w,
                                                                         const float *bias, int relu, float out_scale, float *Y, int ldy) {
    constexpr int N = 16 * NT, LS = K + 4, KB = K / 16;
    extern __shared__ float Wl[];  // [N][LS]
    if (!transpose_w) {
        for (int e = threadIdx.x * 4; e < N * K; e += blockDim.x * 4) {
            const int n = e / K, k = e % K;
            *reinterpret_cast<float4 *>(Wl + n * LS + k) = *reinterpret_cast<const float4 *>(W + e);
        }
    } else {  // W is (K, N): four n of one k per load, stored down a column of the [n][k] image
        for (int e = threadIdx.x * 4; e < N * K; e += blockDim.x * 4) {
            const int k = e / N, n = e % N;
            const float4 v = *reinterpret_cast<const float4 *>(W + e);
            Wl[(n + 0) * LS + k] = v.x; Wl[(n + 1) * LS + k] = v.y; Wl[(n + 2) * LS + k] = v.z; Wl[(n + 3) * LS + k] = v.w;
        }
    }
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    float bcol[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bcol[t] = bias ? bias[16 * t + la] : 0.f;
    __syncthreads();
    const int tiles = (M + 15) / 16, step = gridDim.x * LR_WAVES;
    int tile = blockIdx.x * LR_WAVES + wv;
    float4 xn[KB];
    if (tile < tiles) {
        const float *row = X + (size_t)min(tile * 16 + la, M - 1) * ldx + 4 * g;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) xn[kb] = *reinterpret_cast<const float4 *>(row + 16 * kb);
    }
    for (; tile < tiles; tile += step) {
        float4 xc[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) xc[kb] = xn[kb];
        if (tile + step < tiles) {
            const float *row = X + (size_t)min((tile + step) * 16 + la, M - 1) * ldx + 4 * g;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) xn[kb] = *reinterpret_cast<const float4 *>(row + 16 * kb);
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{bcol[t], bcol[t], bcol[t], bcol[t]};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 b4 = *reinterpret_cast<const float4 *>(Wl + (16 * t + la) * LS + 16 * kb + 4 * g);
                LR_MFMA(acc[t], xc[kb].x, b4.x);
                LR_MFMA(acc[t], xc[kb].y, b4.y);
                LR_MFMA(acc[t], xc[kb].z, b4.z);
                LR_MFMA(acc[t], xc[kb].w, b4.w);
            }
        }
        // lane (la, g) holds Y[16 tile + 4 g + r][16 t + la]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = tile * 16 + 4 * g + r;
            if (m < M) {
                float *out = Y + (size_t)m * ldy + la;
#pragma unroll
                for (int t = 0; t < NT; ++t) out[16 * t] = (relu ? fmaxf(acc[t][r], 0.f) : acc[t][r]) * out_scale;
            }
        }
    }
}

static bool lr_shape(int K, int N) { return (K == 64 || K == 128) && (N == 64 || N == 128); }

extern "C" int mssvt_linear_rows_supported(int K, int N) { return lr_shape(K, N) ? 1 : 0; }

extern "C" int mssvt_linear_rows(int M, int K, int N, const float *X, int ldx, const float *W, int transpose_w, const float *bias,
                                 int relu, float out_scale, float *Y, int ldy, void *stream) {
    if (M < 0 || !X || !W || !Y || ldx < K || ldy < N || (ldx & 3)) return MSSVT_E_BADARG;
    if (!lr_shape(K, N)) return MSSVT_E_TOOLARGE;
    if (M == 0) return MSSVT_OK;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int grid = min(cus, divup(M, 16 * LR_WAVES));
    const size_t lds = (size_t)N * (K + 4) * 4;
#define LR_GO(KK, NN)                                                                                                       \
    if (K == KK && N == NN) {                                                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_linear_rows<KK, NN / 16>),                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
        if (e != hipSuccess) return (int)e;                                                                                 \
        k_linear_rows<KK, NN / 16><<<grid, LR_WAVES * MSSVT_WAVE, lds, st>>>(M, X, ldx, W, transpose_w, bias, relu, out_scale, Y, ldy); \
        return mssvt_launch_status();                                                                                       \
    }
    LR_GO(64, 64) LR_GO(64, 128) LR_GO(128, 64) LR_GO(128, 128)
#undef LR_GO
    return MSSVT_E_TOOLARGE;
}
