// ffn.hip -- fused per-voxel feed-forward tail of an MsSVT block on the fp32 matrix cores.
//
// Replaces, per block, the reference's (ref: mssvt_backbone.py:336-343 / :383-387)
//     x   = features (+ shortcut)           elementwise
//     h   = norm2(x)                        LayerNorm kernel
//     u   = relu(linear1(h))                GEMM (N x C x FF) + bias + ReLU kernels
//     y   = x + linear2(u)                  GEMM (N x FF x C) + bias + add kernels
// and the NEXT block's norm1(y), with ONE kernel that reads x once and writes y (and
// optionally norm1_next(y)) once: 4*C bytes in and 4*C (8*C) bytes out per voxel, the
// N x FF hidden activations never leave the registers.
//
// MFMA mapping (v_mfma_f32_16x16x4_f32, exact fp32; one wavefront = 16 voxel rows):
//   GEMM1 is computed TRANSPOSED:  D1[hidden][row] = sum_c W1[hidden][c] * h[row][c]
//     A = W1 tile from LDS, B = the normalised rows held in registers (lane (row, kk)
//     owns channels [kk*C/4, (kk+1)*C/4) of its row: 128 contiguous bytes from HBM);
//   its accumulator (lane = row, registers = 4 hidden units 4g..4g+3 of the 16-tile)
//   IS the A operand of GEMM2 step by step -- no shuffle, no LDS round trip:
//     D2[row][out] += sum_reg u[row][4g+reg] * W2[out][4g+reg],  B = one ds_read_b128 of W2.
// Weights (2 * C * FF floats = 256 KB at C=128, FF=256) do not fit the 160 KB LDS: they
// stream through it in chunks of 64 hidden units (W1 rows + W2 columns of the chunk,
// ~70 KB), double buffered, loaded global->registers during the previous chunk's MFMAs.
// LDS row strides (C+2 for the b64 reads of W1, 72 for the b128 reads of W2) make both
// operand reads bank-conflict free.
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FFN_WAVES 8
#define FFN_CH 64  // hidden units per LDS chunk

struct FfnArgs {
    int n_rows;
    const float *x_new, *x_in;  // x = owner < 0 ? 2 * x_in : x_new   (x_in/owner may be null)
    const int *owner;
    // interpolation-table input (tab_row != null; replaces x_new/owner):
    //   x = tab_row[v].x < 0 ? 2 * x_in[v] : x_in[v] + sum_i tab_w[v][i] * attn[tab_row[v][i]]
    const int4 *tab_row;
    const float4 *tab_w;
    const float *attn;
    const float *ln_w, *ln_b;  // norm2
    float eps;
    const float *W1, *b1, *W2, *b2;  // linear1 (FF,C), linear2 (C,FF)
    float *y;
    const float *ln2_w, *ln2_b;  // optional second LayerNorm applied to y (next block's norm1)
    float eps2;
    float *y_norm;
};

template <int C, int FF>
__global__ void __launch_bounds__(FFN_WAVES *MSSVT_WAVE) k_ffn(FfnArgs a) {
    constexpr int KS = C / 4;                       // channels per lane slot (= GEMM1 k-steps)
    constexpr int CH = FF < FFN_CH ? FF : FFN_CH;   // hidden units per chunk
    constexpr int NCH = FF / CH;
    constexpr int RS1 = C + 2;                      // W1 chunk row stride (floats)
    constexpr int RS2 = CH + 8;                     // W2 chunk row stride
    constexpr int BUF = CH * RS1 + C * RS2;         // floats per chunk buffer
    constexpr int V1 = CH * C / 2 / (FFN_WAVES * MSSVT_WAVE);  // float2 of W1 per thread per chunk
    constexpr int V2 = C * CH / 4 / (FFN_WAVES * MSSVT_WAVE);  // float4 of W2 per thread per chunk
    static_assert(V1 >= 1 && V2 >= 1, "chunk too small for the staging pattern");
    extern __shared__ float4 lds4[];
    float *lds = reinterpret_cast<float *>(lds4);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;  // (row | out | hidden-in-tile, k slot)
    const int nbatch = (a.n_rows + FFN_WAVES * 16 - 1) / (FFN_WAVES * 16);

    float2 st1[V1];
    float4 st2[V2];
    auto stage_load = [&](int ch) {
#pragma unroll
        for (int v = 0; v < V1; ++v) {  // W1 rows [ch*CH, +CH), all C columns
            const int e = (v * FFN_WAVES * MSSVT_WAVE + tid) * 2;
            st1[v] = *reinterpret_cast<const float2 *>(a.W1 + (size_t)(ch * CH + e / C) * C + e % C);
        }
#pragma unroll
        for (int v = 0; v < V2; ++v) {  // W2 all C rows, columns [ch*CH, +CH)
            const int e = (v * FFN_WAVES * MSSVT_WAVE + tid) * 4;
            st2[v] = *reinterpret_cast<const float4 *>(a.W2 + (size_t)(e / CH) * FF + ch * CH + e % CH);
        }
    };
    auto stage_store = [&](float *buf) {
#pragma unroll
        for (int v = 0; v < V1; ++v) {
            const int e = (v * FFN_WAVES * MSSVT_WAVE + tid) * 2;
            *reinterpret_cast<float2 *>(buf + (e / C) * RS1 + e % C) = st1[v];
        }
#pragma unroll
        for (int v = 0; v < V2; ++v) {
            const int e = (v * FFN_WAVES * MSSVT_WAVE + tid) * 4;
            *reinterpret_cast<float4 *>(buf + CH * RS1 + (e / CH) * RS2 + e % CH) = st2[v];
        }
    };

    stage_load(0);
    stage_store(lds);
    __syncthreads();
    int cur = 0;  // buffer holding the chunk about to be consumed

    for (int batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const int r0 = batch * (FFN_WAVES * 16) + wv * 16;
        // ---- load 16 rows in the B-operand layout, apply norm2 ---------------------------
        const int row = min(r0 + li, a.n_rows - 1);
        float xn[KS];
        {
            if (a.tab_row) {
                // residual input built on the fly: x_in + 3-NN interpolated attention rows (the rows of
                // empty slots point at a zero row, so all three gathers are unconditional)
                const int4 tr = a.tab_row[row];
                const float4 tw = a.tab_w[row];
                const bool unowned = tr.x < 0;
                const float *sx = a.x_in + (size_t)row * C + lg * KS;
                // (an unowned voxel gathers its own finite x_in row three times with weight 0: attn rows
                //  of never-written slots may hold NaNs, and 0 * NaN is NaN)
                const float *s1 = unowned ? sx : a.attn + (size_t)tr.x * C + lg * KS;
                const float *s2 = unowned ? sx : a.attn + (size_t)tr.y * C + lg * KS;
                const float *s3 = unowned ? sx : a.attn + (size_t)tr.z * C + lg * KS;
                const float w1 = unowned ? 0.f : tw.x, w2 = unowned ? 0.f : tw.y, w3 = unowned ? 0.f : tw.z;
                const float wx = unowned ? 2.0f : 1.0f;  // untouched voxel: features + shortcut = 2 * x_in
                float *ys = a.y + (size_t)row * C + lg * KS;
#pragma unroll
                for (int s = 0; s < KS; s += 4) {
                    const float4 vx = *reinterpret_cast<const float4 *>(sx + s);
                    const float4 v1 = *reinterpret_cast<const float4 *>(s1 + s);
                    const float4 v2 = *reinterpret_cast<const float4 *>(s2 + s);
                    const float4 v3 = *reinterpret_cast<const float4 *>(s3 + s);
                    float4 o;
                    o.x = ((v1.x * w1 + v2.x * w2) + v3.x * w3) + vx.x * wx;
                    o.y = ((v1.y * w1 + v2.y * w2) + v3.y * w3) + vx.y * wx;
                    o.z = ((v1.z * w1 + v2.z * w2) + v3.z * w3) + vx.z * wx;
                    o.w = ((v1.w * w1 + v2.w * w2) + v3.w * w3) + vx.w * wx;
                    xn[s] = o.x; xn[s + 1] = o.y; xn[s + 2] = o.z; xn[s + 3] = o.w;
                    // x is parked in the output buffer; the epilogue re-reads it in its own layout
                    if (r0 + li < a.n_rows) *reinterpret_cast<float4 *>(ys + s) = o;
                }
            } else {
            const bool dbl = a.owner != nullptr && a.owner[row] < 0;  // untouched voxel: 2 * x_in (ref quirk R12)
            const float *src = (dbl ? a.x_in : a.x_new) + (size_t)row * C + lg * KS;
#pragma unroll
            for (int s = 0; s < KS; s += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(src + s);
                xn[s] = v.x; xn[s + 1] = v.y; xn[s + 2] = v.z; xn[s + 3] = v.w;
            }
            if (dbl) {
#pragma unroll
                for (int s = 0; s < KS; ++s) xn[s] *= 2.0f;
            }
            }
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) sum += xn[s];
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float mean = sum * (1.0f / C);
            float var = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float d = xn[s] - mean;
                var = __builtin_fmaf(d, d, var);
            }
            var += __shfl_xor(var, 16);
            var += __shfl_xor(var, 32);
            const float rstd = rsqrtf(var * (1.0f / C) + a.eps);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int c = lg * KS + s;
                xn[s] = (xn[s] - mean) * rstd * a.ln_w[c] + a.ln_b[c];
            }
        }
        f32x4 acc2[C / 16];
#pragma unroll
        for (int ot = 0; ot < C / 16; ++ot) acc2[ot] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- hidden dimension in chunks; next chunk's weights are in flight meanwhile ----------
        for (int ch = 0; ch < NCH; ++ch) {
            const float *W1c = lds + cur * BUF;
            const float *W2c = W1c + CH * RS1;
            const int nxt = ch + 1 < NCH ? ch + 1 : 0;
            const bool more = ch + 1 < NCH || batch + (int)gridDim.x < nbatch;
            if (NCH > 1 && more) stage_load(nxt);
#pragma unroll
            for (int ht = 0; ht < CH / 16; ++ht) {
                // GEMM1 (transposed): 16 hidden units x 16 rows, K = C
                f32x4 d = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float *w1 = W1c + (ht * 16 + li) * RS1 + lg * KS;
#pragma unroll
                for (int s2 = 0; s2 < KS / 2; ++s2) {
                    const float2 w = *reinterpret_cast<const float2 *>(w1 + 2 * s2);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, xn[2 * s2], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, xn[2 * s2 + 1], d, 0, 0, 0);
                }
                // bias + ReLU on this lane's 4 hidden units (4g .. 4g+3 of the tile)
                const float4 bb = *reinterpret_cast<const float4 *>(a.b1 + ch * CH + ht * 16 + 4 * lg);
                d[0] = fmaxf(d[0] + bb.x, 0.f);
                d[1] = fmaxf(d[1] + bb.y, 0.f);
                d[2] = fmaxf(d[2] + bb.z, 0.f);
                d[3] = fmaxf(d[3] + bb.w, 0.f);
                // GEMM2: the accumulator is the A operand, one k-step per register
#pragma unroll
                for (int ot = 0; ot < C / 16; ++ot) {
                    const float4 w2 = *reinterpret_cast<const float4 *>(W2c + (ot * 16 + li) * RS2 + ht * 16 + 4 * lg);
                    acc2[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[0], w2.x, acc2[ot], 0, 0, 0);
                    acc2[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[1], w2.y, acc2[ot], 0, 0, 0);
                    acc2[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[2], w2.z, acc2[ot], 0, 0, 0);
                    acc2[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[3], w2.w, acc2[ot], 0, 0, 0);
                }
            }
            if (NCH > 1) {
                if (more) stage_store(lds + (cur ^ 1) * BUF);
                __syncthreads();  // everyone is done with `cur`, and the other buffer is complete
                cur ^= 1;
            }
        }
        // ---- epilogue: y = x + W2 u + b2   (lane = output channel li of each 16-tile, rows 4*lg+reg) --
        // (table mode: the parked x rows were stored long ago -- a whole MFMA phase and at least one
        //  workgroup barrier lie in between; the fence makes the ordering explicit)
        if (a.tab_row) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int rr = r0 + 4 * lg + reg;
            const int r = min(rr, a.n_rows - 1);
            const bool dbl = !a.tab_row && a.owner != nullptr && a.owner[r] < 0;
            const float *src = (a.tab_row ? a.y : (dbl ? a.x_in : a.x_new)) + (size_t)r * C + li;
            float s = 0.f;
#pragma unroll
            for (int ot = 0; ot < C / 16; ++ot) {
                // table mode: x was parked in y by other lanes of this wave -> bypass the (stale) L1
                float xv = a.tab_row ? __builtin_nontemporal_load(src + ot * 16) : src[ot * 16];
                if (dbl) xv *= 2.0f;
                const float v = xv + (acc2[ot][reg] + a.b2[ot * 16 + li]);
                acc2[ot][reg] = v;
                s += v;
            }
            float m = 0.f, rs = 0.f;
            if (a.y_norm) {  // LayerNorm of y for the next block: a row lives in one 16-lane DPP row
                s = row_sum16(s);
                m = s * (1.0f / C);
                float q = 0.f;
#pragma unroll
                for (int ot = 0; ot < C / 16; ++ot) {
                    const float dd = acc2[ot][reg] - m;
                    q = __builtin_fmaf(dd, dd, q);
                }
                q = row_sum16(q);
                rs = rsqrtf(q * (1.0f / C) + a.eps2);
            }
            if (rr < a.n_rows) {
#pragma unroll
                for (int ot = 0; ot < C / 16; ++ot) {
                    const int c = ot * 16 + li;
                    a.y[(size_t)rr * C + c] = acc2[ot][reg];
                    if (a.y_norm) a.y_norm[(size_t)rr * C + c] = (acc2[ot][reg] - m) * rs * a.ln2_w[c] + a.ln2_b[c];
                }
            }
        }
    }
}

template <int C, int FF>
static int launch_ffn(const FfnArgs &a, hipStream_t stream) {
    constexpr int CH = FF < FFN_CH ? FF : FFN_CH;
    constexpr int NCH = FF / CH;
    constexpr int BUF = CH * (C + 2) + C * (CH + 8);
    const size_t lds_bytes = (size_t)BUF * 4 * (NCH > 1 ? 2 : 1);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_ffn<C, FF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    const int nbatch = (a.n_rows + FFN_WAVES * 16 - 1) / (FFN_WAVES * 16);
    const int per_cu = (int)((160 * 1024) / lds_bytes) < 1 ? 1 : (int)((160 * 1024) / lds_bytes);
    int grid = cus * (per_cu > 2 ? 2 : per_cu);
    if (grid > nbatch) grid = nbatch;
    if (grid < 1) return MSSVT_OK;
    k_ffn<C, FF><<<grid, FFN_WAVES * MSSVT_WAVE, lds_bytes, stream>>>(a);
    return mssvt_launch_status();
}

extern "C" int mssvt_ffn_fused(int n_rows, int C, int FF, const float *x_new, const float *x_in,
                               const int *owner, const float *norm_w, const float *norm_b, float eps,
                               const float *W1, const float *b1, const float *W2, const float *b2,
                               float *y, const float *next_norm_w, const float *next_norm_b,
                               float next_eps, float *y_norm, void *stream) {
    if (n_rows < 0 || !x_new || !norm_w || !norm_b || !W1 || !b1 || !W2 || !b2 || !y) return MSSVT_E_BADARG;
    if (owner && !x_in) return MSSVT_E_BADARG;
    if (y_norm && (!next_norm_w || !next_norm_b)) return MSSVT_E_BADARG;
    if (n_rows == 0) return MSSVT_OK;
    FfnArgs a;
    a.n_rows = n_rows; a.x_new = x_new; a.x_in = x_in; a.owner = owner;
    a.ln_w = norm_w; a.ln_b = norm_b; a.eps = eps;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.y = y;
    a.ln2_w = next_norm_w; a.ln2_b = next_norm_b; a.eps2 = next_eps; a.y_norm = y_norm;
    a.tab_row = nullptr; a.tab_w = nullptr; a.attn = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (C == 128 && FF == 256) return launch_ffn<128, 256>(a, st);
    if (C == 64 && FF == 128) return launch_ffn<64, 128>(a, st);
    if (C == 32 && FF == 64) return launch_ffn<32, 64>(a, st);
    return MSSVT_E_TOOLARGE;  // shape not instantiated: callers use library GEMMs instead
}

// Same tail, fed by the interpolation table of mssvt_block_interp_table: the residual input
// x = x_in + sum_i w_i * attn[row_i] (or 2 * x_in for voxels no list slot owns) is built while the
// rows are loaded, so neither the scatter kernel nor its (N,C) output exist.
extern "C" int mssvt_ffn_fused_interp(int n_rows, int C, int FF, const float *x_in, const int *tab_row,
                                      const float *tab_w, const float *attn, const float *norm_w,
                                      const float *norm_b, float eps, const float *W1, const float *b1,
                                      const float *W2, const float *b2, float *y,
                                      const float *next_norm_w, const float *next_norm_b, float next_eps,
                                      float *y_norm, void *stream) {
    if (n_rows < 0 || !x_in || !tab_row || !tab_w || !attn || !norm_w || !norm_b || !W1 || !b1 || !W2 ||
        !b2 || !y)
        return MSSVT_E_BADARG;
    if (y_norm && (!next_norm_w || !next_norm_b)) return MSSVT_E_BADARG;
    if (n_rows == 0) return MSSVT_OK;
    FfnArgs a;
    a.n_rows = n_rows; a.x_new = nullptr; a.x_in = x_in; a.owner = nullptr;
    a.tab_row = reinterpret_cast<const int4 *>(tab_row);
    a.tab_w = reinterpret_cast<const float4 *>(tab_w);
    a.attn = attn;
    a.ln_w = norm_w; a.ln_b = norm_b; a.eps = eps;
    a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.y = y;
    a.ln2_w = next_norm_w; a.ln2_b = next_norm_b; a.eps2 = next_eps; a.y_norm = y_norm;
    hipStream_t st = (hipStream_t)stream;
    if (C == 128 && FF == 256) return launch_ffn<128, 256>(a, st);
    if (C == 64 && FF == 128) return launch_ffn<64, 128>(a, st);
    if (C == 32 && FF == 64) return launch_ffn<32, 64>(a, st);
    return MSSVT_E_TOOLARGE;
}
