// segment_reduce.hip -- deterministic replacement of the reference's atomic scatter-adds in the backward pass.
//
// The reference accumulates gradients of its row gathers with atomicAdd, one thread per gathered element:
//   K6   group_features_grad_kernel_stack   (ref: pcdet/ops/mssvt/src/group_features_gpu.cu:15-47)
//   K11a gather_points_grad_kernel_fast     (ref: pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:53-90)
//   K11b group_points_grad_kernel_fast      (ref: .../group_points_gpu.cu:14-50)
// so the sum order -- and with it the low bits of every gradient -- changes from run to run.  Here the gather is
// inverted ONCE per index set (an "inverted index" in CSR form: for every destination row the list of the
// contribution rows that add into it, in ascending contribution order; mssvt_amd/train_path.py builds it with a
// stable sort) and every destination row is then summed by one group of lanes in that fixed order:
//
//     dst[d][:] = sum_{e in [off[d], off[d+1])}  w[e] * src[idx[e]][:]          (w optional)
//
// No atomics, no dependence on scheduling: bit-identical gradients run to run.  The same kernel is the forward
// of a weighted row gather (3-NN interpolation: three contribution rows per voxel, ref mssvt_backbone.py:298-311)
// and the backward of every plain gather of the training path.  HBM-bound: 4 C bytes per contribution row read
// once (whole rows, 16 bytes per lane), 4 C bytes per destination row written once.
#include "common.hip.h"

template <int LPR>  // lanes per row (power of two <= 64): 64 / LPR destination rows per wavefront
__global__ void __launch_bounds__(256) k_segment_sum_rows(int C, int n_dst, const int *seg_start, const int *seg_end,
                                                          const int *idx, const float *w, const float *src, int src_stride,
                                                          float *dst, int dst_stride, int accumulate, const float *res,
                                                          const float *row_a, const float *row_b) {
    constexpr int RPW = MSSVT_WAVE / LPR;
    const int lane = lane_id(), sub = lane / LPR, l = lane % LPR;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / MSSVT_WAVE;
    const int d = wave * RPW + sub;
    if (d >= n_dst) return;
    const int e0 = seg_start[d], e1 = seg_end[d];
    for (int c = 4 * l; c < C; c += 4 * LPR) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // four contribution rows in flight (indices first, then the rows: two round trips per four entries; most lists
        // of the training path -- keys per voxel, 3-NN rows -- are done in one pass); entries past the end re-read the
        // last one and are not added; the ADD order stays e0, e0 + 1, ... (fixed)
        int e = e0;
        // long lists (the fixed chunks of a heavy destination: 256 entries on one lane group) eight rows at a time
        for (; e + 8 <= e1; e += 8) {
            int r[8];
            float wv[8];
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = idx[e + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w ? w[e + u] : 1.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(src + (size_t)r[u] * src_stride + c);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc.x = __builtin_fmaf(wv[u], v[u].x, acc.x); acc.y = __builtin_fmaf(wv[u], v[u].y, acc.y);
                acc.z = __builtin_fmaf(wv[u], v[u].z, acc.z); acc.w = __builtin_fmaf(wv[u], v[u].w, acc.w);
            }
        }
        for (; e < e1; e += 4) {
            const int last = e1 - 1;
            const int i1 = min(e + 1, last), i2 = min(e + 2, last), i3 = min(e + 3, last);
            const int r0 = idx[e], r1 = idx[i1], r2 = idx[i2], r3 = idx[i3];
            float w0 = 1.0f, w1 = 1.0f, w2 = 1.0f, w3 = 1.0f;
            if (w) { w0 = w[e]; w1 = w[i1]; w2 = w[i2]; w3 = w[i3]; }
            const float4 v0 = *reinterpret_cast<const float4 *>(src + (size_t)r0 * src_stride + c);
            const float4 v1 = *reinterpret_cast<const float4 *>(src + (size_t)r1 * src_stride + c);
            const float4 v2 = *reinterpret_cast<const float4 *>(src + (size_t)r2 * src_stride + c);
            const float4 v3 = *reinterpret_cast<const float4 *>(src + (size_t)r3 * src_stride + c);
            acc.x = __builtin_fmaf(w0, v0.x, acc.x); acc.y = __builtin_fmaf(w0, v0.y, acc.y);
            acc.z = __builtin_fmaf(w0, v0.z, acc.z); acc.w = __builtin_fmaf(w0, v0.w, acc.w);
            if (e + 1 < e1) {
                acc.x = __builtin_fmaf(w1, v1.x, acc.x); acc.y = __builtin_fmaf(w1, v1.y, acc.y);
                acc.z = __builtin_fmaf(w1, v1.z, acc.z); acc.w = __builtin_fmaf(w1, v1.w, acc.w);
            }
            if (e + 2 < e1) {
                acc.x = __builtin_fmaf(w2, v2.x, acc.x); acc.y = __builtin_fmaf(w2, v2.y, acc.y);
                acc.z = __builtin_fmaf(w2, v2.z, acc.z); acc.w = __builtin_fmaf(w2, v2.w, acc.w);
            }
            if (e + 3 < e1) {
                acc.x = __builtin_fmaf(w3, v3.x, acc.x); acc.y = __builtin_fmaf(w3, v3.y, acc.y);
                acc.z = __builtin_fmaf(w3, v3.z, acc.z); acc.w = __builtin_fmaf(w3, v3.w, acc.w);
            }
        }
        float4 *out = reinterpret_cast<float4 *>(dst + (size_t)d * dst_stride + c);
        if (res) {  // dst = row_a[d] * res[d] + row_b[d] * sum: the Block's residual / DropPath epilogue (_residual entry)
            const float4 x = *reinterpret_cast<const float4 *>(res + (size_t)d * C + c);
            const float ra = row_a[d], rb = row_b[d];
            acc.x = __builtin_fmaf(rb, acc.x, ra * x.x); acc.y = __builtin_fmaf(rb, acc.y, ra * x.y);
            acc.z = __builtin_fmaf(rb, acc.z, ra * x.z); acc.w = __builtin_fmaf(rb, acc.w, ra * x.w);
        }
        if (accumulate) {  // dst += the list's sum (one add per element: the order stays fixed)
            const float4 o = *out;
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        *out = acc;
    }
}

// A destination with thousands of contribution rows (the reference's "(x + 0.1).int()" quirk turns every FPS-picked
// empty slot into voxel 0 of the sample: ~one contribution per window) would keep ONE lane group busy for the whole
// launch.  The caller therefore sums such lists in fixed chunks of a few hundred entries (this entry point on
// explicit [start, end) ranges) and adds the chunk sums in chunk order in a second call -- still a fixed order.
extern "C" int mssvt_segment_sum_rows_ranges(int C, int n_dst, const int *seg_start, const int *seg_end,
                                             const int *csr_idx, const float *csr_w, const float *src, float *dst,
                                             void *stream);
extern "C" int mssvt_segment_sum_rows_strided(int C, int n_dst, const int *seg_start, const int *seg_end, const int *csr_idx,
                                              const float *csr_w, const float *src, int src_stride, float *dst, int dst_stride,
                                              int accumulate, void *stream);

extern "C" int mssvt_segment_sum_rows(int C, int n_dst, const int *csr_off, const int *csr_idx, const float *csr_w,
                                      const float *src, float *dst, void *stream) {
    if (!csr_off) return MSSVT_E_BADARG;
    return mssvt_segment_sum_rows_ranges(C, n_dst, csr_off, csr_off + 1, csr_idx, csr_w, src, dst, stream);
}

extern "C" int mssvt_segment_sum_rows_ranges(int C, int n_dst, const int *seg_start, const int *seg_end,
                                             const int *csr_idx, const float *csr_w, const float *src, float *dst,
                                             void *stream) {
    return mssvt_segment_sum_rows_strided(C, n_dst, seg_start, seg_end, csr_idx, csr_w, src, C, dst, C, 0, stream);
}

// The same sum on C columns of wider rows (src / dst row strides in floats, both pointers at the first column), written
// or ADDED to dst: the gradient of a gather of a column range lands in that range of the full-width gradient without a
// zero-filled temporary and an add (the head groups of a Block: train_path._Tokens).
static int seg_launch(int C, int n_dst, const int *seg_start, const int *seg_end, const int *csr_idx, const float *csr_w,
                      const float *src, int src_stride, float *dst, int dst_stride, int accumulate, const float *res,
                      const float *row_a, const float *row_b, void *stream);

extern "C" int mssvt_segment_sum_rows_strided(int C, int n_dst, const int *seg_start, const int *seg_end, const int *csr_idx,
                                              const float *csr_w, const float *src, int src_stride, float *dst, int dst_stride,
                                              int accumulate, void *stream) {
    return seg_launch(C, n_dst, seg_start, seg_end, csr_idx, csr_w, src, src_stride, dst, dst_stride, accumulate, nullptr, nullptr,
                      nullptr, stream);
}

// dst[d] = row_a[d] * res[d] + row_b[d] * (the list's sum): the tail of a Block under autograd in one launch -- 3-NN
// interpolation of the attention rows, the "untouched voxels keep x_in" select, DropPath and the residual add
// (ref mssvt_backbone.py:298-340): row_b = the row's DropPath factor, row_a = 1 for a voxel the attention updates and
// 1 + row_b for one it does not (whose list holds zero weights only).
extern "C" int mssvt_segment_sum_rows_residual(int C, int n_dst, const int *seg_start, const int *seg_end, const int *csr_idx,
                                               const float *csr_w, const float *src, const float *res, const float *row_a,
                                               const float *row_b, float *dst, void *stream) {
    if (!res || !row_a || !row_b) return MSSVT_E_BADARG;
    return seg_launch(C, n_dst, seg_start, seg_end, csr_idx, csr_w, src, C, dst, C, 0, res, row_a, row_b, stream);
}

static int seg_launch(int C, int n_dst, const int *seg_start, const int *seg_end, const int *csr_idx, const float *csr_w,
                      const float *src, int src_stride, float *dst, int dst_stride, int accumulate, const float *res,
                      const float *row_a, const float *row_b, void *stream) {
    const int *csr_off = seg_start;
    if (!seg_start || !seg_end || !csr_idx || !src || !dst || C <= 0 || n_dst < 0) return MSSVT_E_BADARG;
    if ((C & 3) || (src_stride & 3) || (dst_stride & 3) || src_stride < C || dst_stride < C) return MSSVT_E_BADARG;  // 16-byte row pieces
    if (n_dst == 0) return MSSVT_OK;
    hipStream_t st = (hipStream_t)stream;
    const int q = C / 4;
#define SEG_LAUNCH(lpr)                                                                                    \
    {                                                                                                      \
        const int rpw = MSSVT_WAVE / lpr, waves = divup(n_dst, rpw);                                        \
        k_segment_sum_rows<lpr><<<divup(waves, 4), 256, 0, st>>>(C, n_dst, csr_off, seg_end, csr_idx, csr_w, src, src_stride, dst, dst_stride, accumulate, res, row_a, row_b); \
    }
    if (q <= 4) SEG_LAUNCH(4)
    else if (q <= 8) SEG_LAUNCH(8)
    else if (q <= 16) SEG_LAUNCH(16)
    else if (q <= 32) SEG_LAUNCH(32)
    else SEG_LAUNCH(64)
#undef SEG_LAUNCH
    return mssvt_launch_status();
}
