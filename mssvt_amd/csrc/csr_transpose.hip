// csr_transpose.hip -- inverted index of a (weighted) row gather, built on the device without host round trips
// (training path).
//
//   forward   dst[d] = sum_{e in [off[d], off[d+1])} w[e] src[idx[e]]          (off == NULL: one entry per row, e == d)
//   transpose grad_src[s] = sum_{p in [t_off[s], t_off[s+1])} t_w[p] grad_dst[t_idx[p]]
//
// with the contributions of a source row in ASCENDING ENTRY ORDER: the gradient of every gather in the compact training
// path becomes a segmented sum with a fixed summation order (csrc/segment_reduce.hip) instead of the reference's
// atomicAdd scatter (ref group_features_gpu.cu:15-47, sampling_gpu.cu:53-90).  A stable radix sort of (source row, entry)
// pairs (rocPRIM, only the log2(n_src) significant bits), then one pass that finds the segment starts by binary search
// and one that resolves entry -> destination row.  Entries of `drop_src` (the constant zero row) sort behind the last
// real source and are left out of every list.  `max_count` receives the longest list if one exceeds `long_list` entries,
// else 0 (the caller cuts long lists into chunks; it reads the words of all its index sets in one host sync).
#include "common.hip.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

__global__ void __launch_bounds__(256) k_csr_keys(int nnz, int n_src, int drop_src, const int *idx, unsigned int *keys, int *vals,
                                                  int *max_count) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) *max_count = 0;
    if (e >= nnz) return;
    const int s = idx[e];
    keys[e] = (s == drop_src || s < 0 || s >= n_src) ? (unsigned int)n_src : (unsigned int)s;
    vals[e] = e;
}

// t_off[s] = first sorted position whose key is >= s (s = 0 .. n_src); the longest list above `long_list` by atomicMax
__global__ void __launch_bounds__(256) k_csr_offsets(int nnz, int n_src, int long_list, const unsigned int *keys, int *t_off,
                                                     int *max_count) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_src) return;
    auto lower = [&](unsigned int v) {
        int lo = 0, hi = nnz;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < v) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int p = lower((unsigned int)s);
    t_off[s] = p;
    if (s < n_src) {
        const int cnt = lower((unsigned int)s + 1u) - p;
        if (cnt > long_list) atomicMax(max_count, cnt);  // rare: one word for the whole launch
    }
}

__global__ void __launch_bounds__(256) k_csr_fill(int n_keep_max, int n_dst, const int *off, const int *vals, const float *w,
                                                  const int *t_off_last, int *t_idx, float *t_w) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_keep_max || p >= *t_off_last) return;  // positions behind the last real source: dropped entries
    const int e = vals[p];
    int d = e;
    if (off) {  // the destination row whose entry range holds e
        int lo = 0, hi = n_dst - 1;  // the last d with off[d] <= e (empty rows before it share that offset)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (off[mid] <= e) lo = mid; else hi = mid - 1;
        }
        d = lo;
    }
    t_idx[p] = d;
    if (t_w) t_w[p] = w[e];
}

static size_t csr_sort_bytes(int nnz, int bits) {
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs<rocprim::default_config>(nullptr, tmp, (unsigned int *)nullptr, (unsigned int *)nullptr,
                                                             (int *)nullptr, (int *)nullptr, (size_t)nnz, 0u, (unsigned int)bits,
                                                             (hipStream_t)0, false);
    return tmp;
}

static int csr_bits(int n_src) {
    int bits = 1;
    while (bits < 32 && (1ll << bits) <= (long long)n_src) ++bits;  // keys go up to n_src inclusive
    return bits;
}

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

extern "C" long long mssvt_csr_transpose_workspace_bytes(int nnz, int n_src) {
    if (nnz < 0 || n_src < 0) return 0;
    return (long long)(4 * align256((size_t)nnz * 4) + align256(csr_sort_bytes(nnz > 0 ? nnz : 1, csr_bits(n_src))) + 256);
}

extern "C" int mssvt_csr_transpose(int nnz, int n_dst, int n_src, const int *off, const int *idx, const float *w, int drop_src,
                                   int long_list, int *t_off, int *t_idx, float *t_w, int *max_count, void *workspace, void *stream) {
    if (nnz < 0 || n_dst < 0 || n_src < 0 || !t_off || !max_count) return MSSVT_E_BADARG;
    if (nnz > 0 && (!idx || !t_idx || !workspace)) return MSSVT_E_BADARG;
    if ((w == nullptr) != (t_w == nullptr)) return MSSVT_E_BADARG;
    if (!off && n_dst != nnz) return MSSVT_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (nnz == 0) {
        hipError_t e = hipMemsetAsync(t_off, 0, (size_t)(n_src + 1) * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(max_count, 0, 4, st);
        return (int)e;
    }
    char *ws = (char *)workspace;
    const size_t seg = align256((size_t)nnz * 4);
    unsigned int *keys_in = (unsigned int *)ws, *keys_out = (unsigned int *)(ws + seg);
    int *vals_in = (int *)(ws + 2 * seg), *vals_out = (int *)(ws + 3 * seg);
    void *tmp = ws + 4 * seg;
    const int bits = csr_bits(n_src);
    size_t tmp_bytes = csr_sort_bytes(nnz, bits);
    k_csr_keys<<<divup(nnz, 256), 256, 0, st>>>(nnz, n_src, drop_src, idx, keys_in, vals_in, max_count);
    if (rocprim::radix_sort_pairs<rocprim::default_config>(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)nnz, 0u,
                                                           (unsigned int)bits, st, false) != hipSuccess)
        return (int)hipErrorLaunchFailure;
    k_csr_offsets<<<divup(n_src + 1, 256), 256, 0, st>>>(nnz, n_src, long_list, keys_out, t_off, max_count);
    k_csr_fill<<<divup(nnz, 256), 256, 0, st>>>(nnz, n_dst, off, vals_out, w, t_off + n_src, t_idx, t_w);
    return mssvt_launch_status();
}
