// linear_rows_h.hip -- Y = s * act(X B^T + bias) over compact rows with split-fp16 matrix operands, for the LARGE weight
// matrices of the training path (128 <-> 256: linear1 / linear2 of a Block and their input gradients; 64 <-> 128: to_kvs;
// 128 x 128: pos_proj.2 of the CompressBlock -- SURVEY.md section 8 f3; ref nn.Linear in mssvt_backbone.py:339-343,
// mssvt_utils.py:80-83 and their autograd backward).  The framework's library GEMM runs these shapes at 47 - 85 TFLOP/s
// (fp32 instruction; 24 % of the training step's GPU time in round 4); the inference kernel k_ffn_ws sustains 3 x that on
// the same shapes with the arithmetic used here (csrc/ffn.hip): every fp32 operand v as hi = fp16(v), lo = fp16((v - hi) 2^11),
// a product sum = hi hi + 2^-11 (hi lo + lo hi) on v_mfma_f32_16x16x32_f16 with fp32 accumulation.
//
// fp16 has a narrow range and gradients do not respect it, so every ROW of X is normalised by a power of two first
// (s_m = 2^-e(max |X[m][:]|), exact; Y[m][:] depends on X[m][:] only, so the row's results are multiplied by 2^e again --
// also exact): the operand halves always see values in (-2, 2), whatever the scale of the activations or gradients.  The
// weight matrix gets ONE power of two for all its elements the same way (computed while it is staged).
//
// One workgroup of 16 waves per CU: the whole weight matrix sits in LDS as ready B^T fragments (hi and lo images,
// [k step][column tile][lane] x 16 bytes: one conflict-free ds_read_b128 per fragment), staged once and split on the way in
// (`transpose_w`: B[n][k] = W[k][n], the input-gradient form).  A wave streams 16-row tiles: its lanes read the rows as
// 32-byte pieces (lane (m, g) <-> row m, columns 32 P + 8 g ..), split them in registers (the A^T operand of every column
// tile), and produce Y^T[n][m] tile by tile -- the accumulator of the transposed product holds four consecutive n of one
// row, so the epilogue (bias, relu, scales) ends in 16-byte stores.  The next tile's rows are in flight under the
// current tile's products.
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
// waves per workgroup: 16 (4 per SIMD, <= 128 registers each) for K <= 128; 8 for K = 256 (the row's 64 prefetch registers +
// 64 accumulator registers do not fit 128: the 16-wave form spilled 742 registers)
#define LH_WAVES_OF(K) ((K) > 128 ? 8 : 16)
#define LH_SCALE 2048.0f
#define LH_INV (1.0f / 2048.0f)
#define LH_MFMA(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16((av), (bv), acc, 0, 0, 0)

__device__ __forceinline__ void lh_split8(const float4 v0, const float4 v1, float s, h16x8 &hi, h16x8 &lo) {
    const float x[8] = {v0.x * s, v0.y * s, v0.z * s, v0.w * s, v1.x * s, v1.y * s, v1.z * s, v1.w * s};
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const fp16x2 a = __builtin_amdgcn_cvt_pkrtz(x[i], x[i + 1]);
        const fp16x2 c = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)a[0], -LH_SCALE, x[i] * LH_SCALE),
                                                    __builtin_fmaf((float)a[1], -LH_SCALE, x[i + 1] * LH_SCALE));
        hi[i] = (_Float16)a[0]; hi[i + 1] = (_Float16)a[1];
        lo[i] = (_Float16)c[0]; lo[i + 1] = (_Float16)c[1];
    }
}

template <int K, int N>
__global__ void __launch_bounds__(LH_WAVES_OF(K) *MSSVT_WAVE, 1) k_linear_rows_h(int M, const float *X, int ldx, const float *W, int transpose_w,
                                                                           const float *bias, int relu, float out_scale, float *Y,
                                                                           int ldy) {
    constexpr int KS = K / 32, NT = N / 16, IMG = KS * NT * 64;  // h16x8 fragments per (hi or lo) image
    constexpr int LH_WAVES = LH_WAVES_OF(K);
    extern __shared__ float4 lds4[];
    h16x8 *Bh = reinterpret_cast<h16x8 *>(lds4), *Bl = Bh + IMG;
    // stage + split the weights: fragment (P, t, lane = 16 g + n % 16) = B[n = 16 t + lane % 16][k = 32 P + 8 g .. + 8),
    // the whole matrix normalised by ONE power of two (2^-e(max |W|), undone in the epilogue): any finite weight scale works
    constexpr int FR = (IMG + LH_WAVES * MSSVT_WAVE - 1) / (LH_WAVES * MSSVT_WAVE);
    __shared__ float wmax_l[LH_WAVES];
    __shared__ float4 bias_l[N / 4];  // (a global load per column tile inside the row loop is a dependent round trip per tile)
    for (int e = threadIdx.x; e < N / 4; e += blockDim.x)
        bias_l[e] = bias ? reinterpret_cast<const float4 *>(bias)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 wv0[FR], wv1[FR];
    float wmx = 0.f;
#pragma unroll
    for (int i = 0; i < FR; ++i) {
        const int f = threadIdx.x + i * LH_WAVES * MSSVT_WAVE;
        wv0[i] = wv1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < IMG) {
            const int ln = f & 63, t = (f >> 6) % NT, P = (f >> 6) / NT;
            const int n = 16 * t + (ln & 15), k0 = 32 * P + 8 * (ln >> 4);
            if (!transpose_w) {  // W (N, K) row-major
                wv0[i] = *reinterpret_cast<const float4 *>(W + (size_t)n * K + k0);
                wv1[i] = *reinterpret_cast<const float4 *>(W + (size_t)n * K + k0 + 4);
            } else {  // W (K, N) row-major: B[n][k] = W[k][n]
                const float *c = W + (size_t)k0 * N + n;
                wv0[i] = make_float4(c[0], c[N], c[2 * N], c[3 * N]);
                wv1[i] = make_float4(c[4 * N], c[5 * N], c[6 * N], c[7 * N]);
            }
            wmx = fmaxf(wmx, fmaxf(fmaxf(fmaxf(fabsf(wv0[i].x), fabsf(wv0[i].y)), fmaxf(fabsf(wv0[i].z), fabsf(wv0[i].w))),
                                   fmaxf(fmaxf(fabsf(wv1[i].x), fabsf(wv1[i].y)), fmaxf(fabsf(wv1[i].z), fabsf(wv1[i].w)))));
        }
    }
    wmx = wave_max(wmx);
    if (lane_id() == 0) wmax_l[threadIdx.x / MSSVT_WAVE] = wmx;
    __syncthreads();
    wmx = 0.f;
#pragma unroll
    for (int i = 0; i < LH_WAVES; ++i) wmx = fmaxf(wmx, wmax_l[i]);
    const int web = __builtin_bit_cast(int, wmx) & 0x7F800000;
    const bool wnorm = web != 0 && web < 0x7F000000;
    const float w_in = wnorm ? __builtin_bit_cast(float, 0x7F000000 - web) : 1.0f;
    const float w_un = wnorm ? __builtin_bit_cast(float, web) : 1.0f;
#pragma unroll
    for (int i = 0; i < FR; ++i) {
        const int f = threadIdx.x + i * LH_WAVES * MSSVT_WAVE;
        if (f < IMG) {
            h16x8 h, l;
            lh_split8(wv0[i], wv1[i], w_in, h, l);
            Bh[f] = h;
            Bl[f] = l;
        }
    }
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    __syncthreads();
    const int tiles = (M + 15) / 16, step = gridDim.x * LH_WAVES;
    int tile = blockIdx.x * LH_WAVES + wv;
    float4 xn[KS][2];
    if (tile < tiles) {
        const float *row = X + (size_t)min(tile * 16 + la, M - 1) * ldx + 8 * g;
#pragma unroll
        for (int P = 0; P < KS; ++P) {
            xn[P][0] = *reinterpret_cast<const float4 *>(row + 32 * P);
            xn[P][1] = *reinterpret_cast<const float4 *>(row + 32 * P + 4);
        }
    }
    for (; tile < tiles; tile += step) {
        // the row's power-of-two normalisation: s = 2^(127 - E), E = biased exponent of max |x| over the row (all four g lanes)
        float mx = 0.f;
#pragma unroll
        for (int P = 0; P < KS; ++P)
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(xn[P][0].x), fabsf(xn[P][0].y)), fmaxf(fabsf(xn[P][0].z), fabsf(xn[P][0].w))),
                                 fmaxf(fmaxf(fabsf(xn[P][1].x), fabsf(xn[P][1].y)), fmaxf(fabsf(xn[P][1].z), fabsf(xn[P][1].w)))));
        mx = fmaxf(mx, lane_xor16(mx));
        mx = fmaxf(mx, lane_xor32(mx));
        const int eb = __builtin_bit_cast(int, mx) & 0x7F800000;
        const bool norm = eb != 0 && eb < 0x7F000000;  // zero / denormal rows and inf / nan rows pass through unscaled
        const float s_in = norm ? __builtin_bit_cast(float, 0x7F000000 - eb) : 1.0f;
        const float un = (norm ? __builtin_bit_cast(float, eb) : 1.0f) * w_un;  // both normalisations undone: exact powers of two
        const int m = tile * 16 + la;
        float *out = Y + (size_t)min(m, M - 1) * ldy + 4 * g;
        // epilogue of column tile t: lane (m = la, g) holds Y[m][16 t + 4 g + i]
        auto finish = [&](int t, const f32x4 &mm, const f32x4 &cr) {
            const float4 b4 = bias_l[4 * t + g];
            float r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(cr[i], LH_INV, mm[i]);
            // (x s_in) (W w_in)^T / (s_in w_in): the products come back at their own scale before the bias joins them
            float4 o;
            o.x = r[0] * un + b4.x; o.y = r[1] * un + b4.y; o.z = r[2] * un + b4.z; o.w = r[3] * un + b4.w;
            if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            o.x *= out_scale; o.y *= out_scale; o.z *= out_scale; o.w *= out_scale;
            if (m < M) *reinterpret_cast<float4 *>(out + 16 * t) = o;
        };
        if constexpr (KS <= 4) {
            // few k steps: the row's fragments stay in registers (8 KS), column tiles one after the other
            h16x8 ah[KS], al[KS];
#pragma unroll
            for (int P = 0; P < KS; ++P) lh_split8(xn[P][0], xn[P][1], s_in, ah[P], al[P]);
            if (tile + step < tiles) {
                const float *row = X + (size_t)min((tile + step) * 16 + la, M - 1) * ldx + 8 * g;
#pragma unroll
                for (int P = 0; P < KS; ++P) {
                    xn[P][0] = *reinterpret_cast<const float4 *>(row + 32 * P);
                    xn[P][1] = *reinterpret_cast<const float4 *>(row + 32 * P + 4);
                }
            }
#pragma unroll 2
            for (int t = 0; t < NT; ++t) {
                f32x4 mm = f32x4{0.f, 0.f, 0.f, 0.f}, cr = mm;
#pragma unroll
                for (int P = 0; P < KS; ++P) {
                    const h16x8 bh = Bh[(P * NT + t) * 64 + lane], bl = Bl[(P * NT + t) * 64 + lane];
                    LH_MFMA(mm, bh, ah[P]);  // Y^T tile: A = the weights' rows n, B = the tile's rows m
                    LH_MFMA(cr, bh, al[P]);
                    LH_MFMA(cr, bl, ah[P]);
                }
                finish(t, mm, cr);
            }
        } else {
            // many k steps, few column tiles (256 -> 128): every column tile's sums stay in registers (8 NT), the row's
            // fragments are split one k step at a time (8 live registers instead of 8 KS)
            static_assert(NT <= 8, "accumulators of all column tiles in registers");
            f32x4 mm[NT], cr[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) mm[t] = cr[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int P = 0; P < KS; ++P) {
                h16x8 ah, al;
                lh_split8(xn[P][0], xn[P][1], s_in, ah, al);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const h16x8 bh = Bh[(P * NT + t) * 64 + lane], bl = Bl[(P * NT + t) * 64 + lane];
                    LH_MFMA(mm[t], bh, ah);
                    LH_MFMA(cr[t], bh, al);
                    LH_MFMA(cr[t], bl, ah);
                }
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler would hoist every fragment read of the unrolled loop: spills)
            }
            if (tile + step < tiles) {
                const float *row = X + (size_t)min((tile + step) * 16 + la, M - 1) * ldx + 8 * g;
#pragma unroll
                for (int P = 0; P < KS; ++P) {
                    xn[P][0] = *reinterpret_cast<const float4 *>(row + 32 * P);
                    xn[P][1] = *reinterpret_cast<const float4 *>(row + 32 * P + 4);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) finish(t, mm[t], cr[t]);
        }
    }
}

static bool lh_shape(int K, int N) {
    return (K == 128 && N == 256) || (K == 256 && N == 128) || (K == 64 && N == 128) || (K == 128 && N == 64) ||
           (K == 128 && N == 128) || (K == 64 && N == 64);
}

extern "C" int mssvt_linear_rows_h_supported(int K, int N) { return lh_shape(K, N) ? 1 : 0; }

// X (M, ldx >= K) f32, W (N, K) -- or (K, N) with transpose_w -- f32, bias (N) or NULL,
// Y (M, ldy >= N) f32 = out_scale * act(X B^T + bias); ldx, ldy multiples of 4.
extern "C" int mssvt_linear_rows_h(int M, int K, int N, const float *X, int ldx, const float *W, int transpose_w, const float *bias,
                                   int relu, float out_scale, float *Y, int ldy, void *stream) {
    if (M < 0 || !X || !W || !Y || ldx < K || ldy < N || (ldx & 3) || (ldy & 3) || out_scale == 0.f) return MSSVT_E_BADARG;
    if (!lh_shape(K, N)) return MSSVT_E_TOOLARGE;
    if (M == 0) return MSSVT_OK;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int grid = min(cus, divup(M, 16 * LH_WAVES_OF(K)));
    const size_t lds = (size_t)K * N * 4;  // two fp16 images
#define LH_GO(KK, NN)                                                                                                    \
    if (K == KK && N == NN) {                                                                                            \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_linear_rows_h<KK, NN>),                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
        if (e != hipSuccess) return (int)e;                                                                              \
        k_linear_rows_h<KK, NN><<<grid, LH_WAVES_OF(KK) * MSSVT_WAVE, lds, st>>>(M, X, ldx, W, transpose_w, bias, relu, out_scale, Y, ldy); \
        return mssvt_launch_status();                                                                                    \
    }
    LH_GO(128, 256) LH_GO(256, 128) LH_GO(64, 128) LH_GO(128, 64) LH_GO(128, 128) LH_GO(64, 64)
#undef LH_GO
    return MSSVT_E_TOOLARGE;
}
