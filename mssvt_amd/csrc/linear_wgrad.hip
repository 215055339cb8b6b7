// linear_wgrad.hip -- weight / bias gradient of an nn.Linear over compact rows, deterministic split-K on the fp32
// matrix cores (training path).
//
//     dW[o][c] = sum_m dY[m][o] * X[m][c]          db[o] = sum_m dY[m][o]          (M rows: 30k .. 600k; Cout, Cin <= 256)
//
// The reduction dimension is the ROW count and the output is at most 256 x 128 -- the shape library GEMMs handle worst
// (no split over K: 64 output tiles on 256 CUs; 435 us per call at 74k rows where the matrix pipe needs 31 us).  Here
// every workgroup takes a contiguous slice of the rows, stages 16 rows of X and dY at a time through LDS (coalesced
// 16-byte loads; the MFMA operands are then single-dword LDS reads, [row][column] with the column on the lane: conflict
// free) and keeps the WHOLE Cout x Cin partial in accumulators (v_mfma_f32_16x16x4_f32: D[o][c] += dY[m][o] X[m][c], four
// rows per instruction; the tile list of a wave is a runtime table, its accumulators a compile-time array).  A bias
// gradient rides along as one more column tile against a constant 1.  The slices' partial slabs are then added in slice
// order by a second launch: no atomics, bit-identical from run to run.
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define WG_WAVES 4
#define WG_ROWS 16  // rows staged per step
#define MFMA4(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((av), (bv), acc, 0, 0, 0)

// slab layout: [slice][tile][lane 64][4] floats, tile = to * ntc_ext + tc (tc == ntc: the bias column tile)
template <int TPW>
__global__ void __launch_bounds__(WG_WAVES *MSSVT_WAVE) k_wgrad_partial(int M, int Cin, int Cout, const float *X, const float *dY,
                                                                         int rows_per_slice, int with_bias, float *slab) {
    extern __shared__ float lds[];
    const int nto = (Cout + 15) / 16, ntc = (Cin + 15) / 16, ntc_ext = ntc + (with_bias ? 1 : 0);
    const int CinP = ntc * 16, CoutP = nto * 16;
    float *Xl = lds;                    // [WG_ROWS][CinP]
    float *Yl = lds + WG_ROWS * CinP;   // [WG_ROWS][CoutP]
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    const int tiles = nto * ntc_ext;
    f32x4 acc[TPW];
    int aoff[TPW], boff[TPW];  // wave-uniform LDS column offsets of the tile's operands (boff < 0: the bias column)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int tile = wv * TPW + t;
        aoff[t] = 16 * (tile / ntc_ext);
        boff[t] = (tile % ntc_ext) < ntc ? 16 * (tile % ntc_ext) : -1;
    }
    const int m0 = blockIdx.x * rows_per_slice, m1 = min(M, m0 + rows_per_slice);
    for (int mb = m0; mb < m1; mb += WG_ROWS) {
        // stage WG_ROWS rows (zero rows past the slice: they add nothing)
        for (int e = threadIdx.x * 4; e < WG_ROWS * CinP; e += blockDim.x * 4) {
            const int r = e / CinP, c = e % CinP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mb + r < m1 && c < Cin) v = *reinterpret_cast<const float4 *>(X + (size_t)(mb + r) * Cin + c);
            *reinterpret_cast<float4 *>(Xl + e) = v;
        }
        for (int e = threadIdx.x * 4; e < WG_ROWS * CoutP; e += blockDim.x * 4) {
            const int r = e / CoutP, c = e % CoutP;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mb + r < m1 && c < Cout) v = *reinterpret_cast<const float4 *>(dY + (size_t)(mb + r) * Cout + c);
            *reinterpret_cast<float4 *>(Yl + e) = v;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < WG_ROWS / 4; ++s) {
            const int row = 4 * s + g;  // k index of this lane inside the 4-row step
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tile = wv * TPW + t;
                if (tile < tiles) {  // wave-uniform
                    const float av = Yl[row * CoutP + aoff[t] + la];
                    const float bv = boff[t] >= 0 ? Xl[row * CinP + boff[t] + la] : 1.0f;
                    MFMA4(acc[t], av, bv);
                }
            }
        }
        __syncthreads();
    }
    float *out = slab + (size_t)blockIdx.x * tiles * 256;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wv * TPW + t;
        if (tile < tiles) *reinterpret_cast<f32x4 *>(out + ((size_t)tile * 64 + lane) * 4) = acc[t];
    }
}

// The shapes of the benchmark configuration: Cout == 64 * OW (wave w owns output tiles [w OW, (w+1) OW)), Cin == 16 * NTC.
// Operand fragments are read from LDS once per 4-row step and reused across the wave's OW x (NTC + 1) tiles; the next
// 16 rows are in flight (registers) while the current ones are multiplied; two LDS buffers, one barrier per step.
template <int OW, int NTC>
__global__ void __launch_bounds__(WG_WAVES *MSSVT_WAVE) k_wgrad_tiled(int M, const float *X, const float *dY, int rows_per_slice,
                                                                       float *slab) {
    constexpr int Cin = 16 * NTC, Cout = 64 * OW, NT = 256;
    constexpr int XQ = WG_ROWS * Cin / 4 / NT, YQ = WG_ROWS * Cout / 4 / NT;  // float4 pieces per thread and step
    static_assert(WG_ROWS * Cin / 4 % NT == 0 && WG_ROWS * Cout / 4 % NT == 0, "stage divides over the workgroup");
    extern __shared__ float lds[];
    constexpr int BUF = WG_ROWS * (Cin + Cout);
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    f32x4 acc[OW][NTC + 1];
#pragma unroll
    for (int o = 0; o < OW; ++o)
#pragma unroll
        for (int c = 0; c <= NTC; ++c) acc[o][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m0 = blockIdx.x * rows_per_slice, m1 = min(M, m0 + rows_per_slice);
    float4 xr[XQ], yr[YQ];
    auto fetch = [&](int mb) {
#pragma unroll
        for (int i = 0; i < XQ; ++i) {
            const int e = (i * NT + threadIdx.x) * 4, r = e / Cin, c = e % Cin;
            xr[i] = mb + r < m1 ? *reinterpret_cast<const float4 *>(X + (size_t)(mb + r) * Cin + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < YQ; ++i) {
            const int e = (i * NT + threadIdx.x) * 4, r = e / Cout, c = e % Cout;
            yr[i] = mb + r < m1 ? *reinterpret_cast<const float4 *>(dY + (size_t)(mb + r) * Cout + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(m0);
    int buf = 0;
    for (int mb = m0; mb < m1; mb += WG_ROWS, buf ^= 1) {
        float *Xl = lds + buf * BUF, *Yl = Xl + WG_ROWS * Cin;
#pragma unroll
        for (int i = 0; i < XQ; ++i) *reinterpret_cast<float4 *>(Xl + (i * NT + threadIdx.x) * 4) = xr[i];
#pragma unroll
        for (int i = 0; i < YQ; ++i) *reinterpret_cast<float4 *>(Yl + (i * NT + threadIdx.x) * 4) = yr[i];
        __syncthreads();  // the other buffer was last read before the previous barrier
        if (mb + WG_ROWS < m1) fetch(mb + WG_ROWS);
#pragma unroll
        for (int s = 0; s < WG_ROWS / 4; ++s) {
            const int row = 4 * s + g;
            float av[OW], bv[NTC];
#pragma unroll
            for (int o = 0; o < OW; ++o) av[o] = Yl[row * Cout + 16 * (wv * OW + o) + la];
#pragma unroll
            for (int c = 0; c < NTC; ++c) bv[c] = Xl[row * Cin + 16 * c + la];
#pragma unroll
            for (int o = 0; o < OW; ++o) {
#pragma unroll
                for (int c = 0; c < NTC; ++c) MFMA4(acc[o][c], av[o], bv[c]);
                MFMA4(acc[o][NTC], av[o], 1.0f);
            }
        }
    }
    float *out = slab + (size_t)blockIdx.x * (4 * OW * (NTC + 1)) * 256;
#pragma unroll
    for (int o = 0; o < OW; ++o)
#pragma unroll
        for (int c = 0; c <= NTC; ++c)
            *reinterpret_cast<f32x4 *>(out + ((size_t)((wv * OW + o) * (NTC + 1) + c) * 64 + lane) * 4) = acc[o][c];
}

// dW / db = sum over the slices, in slice order; one thread per accumulator quad
__global__ void __launch_bounds__(256) k_wgrad_reduce(int slices, int Cin, int Cout, int with_bias, const float *slab, float *dW,
                                                      float *db) {
    const int nto = (Cout + 15) / 16, ntc = (Cin + 15) / 16, ntc_ext = ntc + (with_bias ? 1 : 0);
    const int tiles = nto * ntc_ext;
    // 32 accumulator quads per workgroup x 8 groups of slices (slice s in group s % 8), the 8 partial sums added in group order
    __shared__ f32x4 part[8][32];
    const int ql = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int q = blockIdx.x * 32 + ql;  // (tile, lane)
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < tiles * 64)
#pragma unroll 4  // the loads of four slices in flight, added in slice order
        for (int s = grp; s < slices; s += 8) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(slab + ((size_t)s * tiles * 64 + q) * 4);
            sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
        }
    part[grp][ql] = sum;
    __syncthreads();
    if (grp != 0 || q >= tiles * 64) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const f32x4 v = part[k][ql];
        sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
    }
    const int tile = q / 64, lane = q % 64, la = lane & 15, g = lane >> 4;
    const int to = tile / ntc_ext, tc = tile % ntc_ext;
    // accumulator layout: lane (column la, g) holds rows 4 g + i of the 16 x 16 tile: row = output o, column = input c
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = 16 * to + 4 * g + i;
        if (o >= Cout) continue;
        if (tc < ntc) {
            const int c = 16 * tc + la;
            if (c < Cin) dW[(size_t)o * Cin + c] = sum[i];
        } else if (la == 0 && db) {
            db[o] = sum[i];
        }
    }
}

extern "C" long long mssvt_linear_wgrad_workspace_floats(int M, int Cin, int Cout) {
    const long long nto = (Cout + 15) / 16, ntc = (Cin + 15) / 16 + 1;
    return 1024LL * nto * ntc * 256;  // at most 1024 slices
}

extern "C" int mssvt_linear_wgrad(int M, int Cin, int Cout, const float *X, const float *dY, float *dW, float *db,
                                  float *workspace, void *stream) {
    if (M < 0 || Cin <= 0 || Cout <= 0 || !X || !dY || !dW || !workspace) return MSSVT_E_BADARG;
    if ((Cin & 3) || (Cout & 3)) return MSSVT_E_BADARG;  // 16-byte row pieces
    const int nto = (Cout + 15) / 16, ntc = (Cin + 15) / 16, with_bias = db ? 1 : 0;
    const int tiles = nto * (ntc + with_bias);
    if (tiles > WG_WAVES * 40) return MSSVT_E_TOOLARGE;  // 160 tiles: 256 x 128 and 128 x 256 with their bias columns
    hipStream_t st = (hipStream_t)stream;
    if (M == 0) {
        hipError_t e = hipMemsetAsync(dW, 0, (size_t)Cout * Cin * 4, st);
        if (e == hipSuccess && db) e = hipMemsetAsync(db, 0, (size_t)Cout * 4, st);
        return (int)e;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    // slices: one per CU, two once a slice still gets >= 1024 rows (measured: at 74k rows one slice per CU beats two, 78 vs
    // 83 us -- the slab write + ordered sum is the fixed cost -- and 291 slices on 256 CUs cost 97 us: a second round; at
    // 600k rows two per CU win, 427 vs 451 us); each a multiple of the 16-row staging step
    int slices = (M >= cus * 2 * 1024) ? cus * 2 : cus;
    int rps = (M + slices - 1) / slices;
    rps = (rps + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
    if (rps < 64) rps = 64;
    slices = (M + rps - 1) / rps;
    const size_t lds = (size_t)WG_ROWS * (nto * 16 + ntc * 16) * 4;
    const int tpw = (tiles + WG_WAVES - 1) / WG_WAVES;
    // the benchmark configuration's shapes: fragments reused in registers, double-buffered staging (bias always computed)
#define WG_TILED(OW, NTC)                                                                                              \
    if (Cout == 64 * OW && Cin == 16 * NTC) {                                                                          \
        k_wgrad_tiled<OW, NTC><<<slices, WG_WAVES * MSSVT_WAVE, 2 * lds, st>>>(M, X, dY, rps, workspace);               \
        k_wgrad_reduce<<<divup(4 * OW * (NTC + 1) * 64, 32), 256, 0, st>>>(slices, Cin, Cout, 1, workspace, dW, db);    \
        return mssvt_launch_status();                                                                                  \
    }
    WG_TILED(4, 8) WG_TILED(2, 16) WG_TILED(2, 8) WG_TILED(2, 4) WG_TILED(1, 4) WG_TILED(1, 8) WG_TILED(4, 4)
#undef WG_TILED
#define WG_LAUNCH(T) k_wgrad_partial<T><<<slices, WG_WAVES * MSSVT_WAVE, lds, st>>>(M, Cin, Cout, X, dY, rps, with_bias, workspace)
    if (tpw <= 1) WG_LAUNCH(1);
    else if (tpw <= 2) WG_LAUNCH(2);
    else if (tpw <= 4) WG_LAUNCH(4);
    else if (tpw <= 8) WG_LAUNCH(8);
    else if (tpw <= 16) WG_LAUNCH(16);
    else if (tpw <= 32) WG_LAUNCH(32);
    else WG_LAUNCH(40);
#undef WG_LAUNCH
    k_wgrad_reduce<<<divup(tiles * 64, 32), 256, 0, st>>>(slices, Cin, Cout, with_bias, workspace, dW, db);
    return mssvt_launch_status();
}
