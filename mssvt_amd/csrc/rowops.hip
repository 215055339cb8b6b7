// rowops.hip -- small per-row helpers of the fused path that would otherwise be framework launches.
//
//   mssvt_batch_counts : rows per sample of a (N,4) [b,z,y,x] index tensor.  The reference loops over
//                        the samples on the host with .item() (ref: mssvt_utils.py:35-37,
//                        mssvt_backbone.py:124-130); a framework bincount costs two reductions
//                        (min / max), a histogram and a host synchronisation.
//   mssvt_layer_norm   : norm1 of the first block (ref: mssvt_backbone.py:241; every later norm1 is
//                        emitted by the previous block's fused FFN epilogue, csrc/ffn.hip).
#include "common.hip.h"

#define BC_ROWS 4    // rows per thread (16: 19 workgroups for a 74k-voxel scene, 10.9 us of serial row loads; 4: 73 workgroups)
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

// LPR lanes (float4 each) per row, 64 / LPR rows per wavefront instruction (the row body lives in common.hip.h: the frame
// call runs it inside its fill launch)
template <int LPR>
__global__ void __launch_bounds__(256) k_layer_norm(const float *x, int n, const float *w, const float *b, float eps,
                                                    float *y) {
    layer_norm_rows<LPR>(x, n, w, b, eps, y, blockIdx.x);
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

// ---- LayerNorm backward (training path): dx per row, d(weight) / d(bias) as deterministic column sums -----------------
// Same row layout as k_layer_norm (LPR lanes x 4 channels per row).  A workgroup walks a contiguous chunk of rows and
// keeps its column sums in registers; lanes / waves holding the same channels are then added in a fixed order through
// LDS and the workgroup's partial row goes to `part` ([workgroup][2][C]); k_layer_norm_bwd_reduce adds the partial rows in
// workgroup order.  No atomics: bit-identical from run to run (torch's kernel is, too; this one reads x and dy once and
// recomputes mean / rstd instead of keeping them: 12 C bytes per row against 20 C).
#define LNB_WG 512  // workgroups (partial rows)
template <int LPR>
__global__ void __launch_bounds__(256) k_layer_norm_bwd(const float *x, const float *dy, int n, const float *w, float eps,
                                                        int rows_per_wg, float *dx, float *part, const float *dres) {
    constexpr int C = LPR * 4, RPW = MSSVT_WAVE / LPR, RPI = 4 * RPW;  // rows per workgroup instruction
    __shared__ float red[4 * RPW][2 * C];
    const int lane = lane_id(), wv = threadIdx.x / MSSVT_WAVE;
    const int rl = wv * RPW + lane / LPR, col = (lane % LPR) * 4;
    const float4 g4 = *reinterpret_cast<const float4 *>(w + col);
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg;
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(n, r0 + rows_per_wg);
    for (int rb = r0; rb < r1; rb += RPI) {
        const int row = rb + rl;
        const bool live = row < r1;
        const size_t off = (size_t)(live ? row : r0) * C + col;
        const float4 v = *reinterpret_cast<const float4 *>(x + off);
        float4 d = *reinterpret_cast<const float4 *>(dy + off);
        if (!live) d = make_float4(0.f, 0.f, 0.f, 0.f);
        float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
        for (int o = LPR / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        const float m = s * (1.0f / C);
        const float cx = v.x - m, cy = v.y - m, cz = v.z - m, cw = v.w - m;
        float q = (cx * cx + cy * cy) + (cz * cz + cw * cw);
#pragma unroll
        for (int o = LPR / 2; o >= 1; o >>= 1) q += __shfl_xor(q, o);
        const float rs = rsqrtf(q * (1.0f / C) + eps);
        const float hx = cx * rs, hy = cy * rs, hz = cz * rs, hw = cw * rs;  // normalised row
        const float gx = d.x * g4.x, gy = d.y * g4.y, gz = d.z * g4.z, gw = d.w * g4.w;
        float a = (gx + gy) + (gz + gw), b = (gx * hx + gy * hy) + (gz * hz + gw * hw);
#pragma unroll
        for (int o = LPR / 2; o >= 1; o >>= 1) {
            a += __shfl_xor(a, o);
            b += __shfl_xor(b, o);
        }
        a *= (1.0f / C);
        b *= (1.0f / C);
        if (live) {
            float4 o = make_float4(rs * (gx - a - hx * b), rs * (gy - a - hy * b), rs * (gz - a - hz * b), rs * (gw - a - hw * b));
            if (dres) {  // + the gradient that reaches x past the LayerNorm (its residual use): one add, no extra pass
                const float4 r = *reinterpret_cast<const float4 *>(dres + off);
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            *reinterpret_cast<float4 *>(dx + off) = o;
        }
        sg.x += d.x * hx; sg.y += d.y * hy; sg.z += d.z * hz; sg.w += d.w * hw;
        sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
    }
    *reinterpret_cast<float4 *>(&red[rl][col]) = sg;
    *reinterpret_cast<float4 *>(&red[rl][C + col]) = sb;
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 4 * RPW; ++k) t += red[k][c];
        part[(size_t)blockIdx.x * 2 * C + c] = t;
    }
}

// one WAVE per column: lane l adds the partial rows l, l + 64, ... in ascending order, then a fixed reduction tree over the
// lanes -- deterministic, and 8 dependent loads per lane instead of 512 per thread (one thread per column took 112 us per
// call: 1.1 ms of a 160k-point training step)
__global__ void __launch_bounds__(256) k_layer_norm_bwd_reduce(const float *part, int nwg, int C2, float *dw, float *db) {
    const int c = blockIdx.x * (blockDim.x / MSSVT_WAVE) + threadIdx.x / MSSVT_WAVE, lane = lane_id();
    if (c >= C2) return;
    float t = 0.f;
    for (int k = lane; k < nwg; k += MSSVT_WAVE) t += part[(size_t)k * C2 + c];
    t = wave_sum(t);
    if (lane == 0) {
        if (c < C2 / 2) dw[c] = t;
        else db[c - C2 / 2] = t;
    }
}

extern "C" int mssvt_layer_norm_backward_residual(const float *x, const float *dy, const float *dres, int num_rows, int C,
                                                  const float *weight, float eps, float *dx, float *dweight, float *dbias,
                                                  float *workspace, void *stream_);

extern "C" int mssvt_layer_norm_backward(const float *x, const float *dy, int num_rows, int C, const float *weight, float eps,
                                         float *dx, float *dweight, float *dbias, float *workspace, void *stream_) {
    return mssvt_layer_norm_backward_residual(x, dy, nullptr, num_rows, C, weight, eps, dx, dweight, dbias, workspace, stream_);
}

// dx = (LayerNorm backward of dy) + dres: `dres` (num_rows, C) or NULL = the gradient that reaches x through its other
// uses (the residual connection around the normalised branch: ref mssvt_backbone.py:241, :339-343)
extern "C" int mssvt_layer_norm_backward_residual(const float *x, const float *dy, const float *dres, int num_rows, int C,
                                                  const float *weight, float eps, float *dx, float *dweight, float *dbias,
                                                  float *workspace, void *stream_) {
    if (!x || !dy || !weight || !dx || !dweight || !dbias || !workspace || num_rows < 0 || C <= 0) return MSSVT_E_BADARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (num_rows == 0) {
        hipError_t e = hipMemsetAsync(dweight, 0, (size_t)C * 4, stream);
        if (e == hipSuccess) e = hipMemsetAsync(dbias, 0, (size_t)C * 4, stream);
        return (int)e;
    }
#define LNB_CASE(lpr)                                                                                        \
    if (C == 4 * lpr) {                                                                                      \
        const int rpi = 4 * (MSSVT_WAVE / lpr);                                                              \
        int rows = (num_rows + LNB_WG - 1) / LNB_WG;                                                         \
        rows = (rows + rpi - 1) / rpi * rpi;                                                                 \
        const int nwg = (num_rows + rows - 1) / rows;                                                        \
        k_layer_norm_bwd<lpr><<<nwg, 256, 0, stream>>>(x, dy, num_rows, weight, eps, rows, dx, workspace, dres); \
        k_layer_norm_bwd_reduce<<<divup(2 * C, 4), 256, 0, stream>>>(workspace, nwg, 2 * C, dweight, dbias);  \
        return mssvt_launch_status();                                                                        \
    }
    LNB_CASE(4)
    LNB_CASE(8)
    LNB_CASE(16)
    LNB_CASE(32)
    LNB_CASE(64)
#undef LNB_CASE
    return MSSVT_E_TOOLARGE;
}
