// pfn_sorted.hip -- DynamicVFE's reductions and PFN layers (eval mode, default configuration) over points GROUPED BY VOXEL:
// no floating-point or integer-view atomics on feature rows, no -inf fills, the per-point layer-2 output never reaches memory.
//
//     f    = [x, y, z, i, e,  xyz - mean_xyz[voxel],  xyz - centre(voxel)]                               (P, 11)
//     x1   = relu(bn1(W1 f + b1));           m1  = max over the voxel's points of x1                     (P, 64) / (N, 64)
//     x2   = relu(bn2(W2 [x1 ; m1[voxel]] + b2));   out = max over the voxel's points of x2              (N, 128)
// (ref pcdet/models/backbones_3d/vfe/dynamic_vfe.py:96-131, PFNLayerV2 :14-52; torch_scatter's scatter_mean / scatter_max.)
//
// csrc/pfn_fused.hip (round 5) reduced x1 / x2 with one atomic per point and channel -- 160k x 192 atomics per frame run at
// the memory-side atomic units' rate (57 + 32 us), on -inf-filled outputs (13 us), after a scatter_mean on 64-bit atomics
// (34 us), and x2 (P, 128) made a round trip through HBM (164 MB).  Here the points are first grouped by voxel with a
// counting sort on the voxel index the voxelizer already produced (the voxels are few-point runs: 2.1 points per voxel at
// 160k points), after which every reduction is a loop over a run:
//   k_ps_rank    per point: slot = atomicAdd(count[voxel], 1)                  (one returning integer atomic per point)
//   k_ps_scan1/2 exclusive scan of the counts (two levels)
//   k_ps_place   the point's row moves to place start[voxel] + slot of the sorted order, row_voxel[...] = voxel; start[]; zero rows
//                for the voxels a layer-2 tile boundary cuts
//   k_ps_mean    16 lanes per voxel: the cluster centre from exact 64-bit fixed-point sums (the arithmetic of csrc/vfe.hip)
//   k_ps_pfn1    8 lanes per SORTED ROW: layer 1 (x1 rows written in sorted order: a voxel's rows are contiguous)
//   k_ps_max1    16 lanes per voxel: m1 = max over the voxel's run of x1 rows
//                (one kernel with 16 lanes per VOXEL doing all three was measured first: 81 us -- a chain of dependent loads per
//                point of the run; split like this every load of a thread is independent of its others)
//   k_ps_pfn2    the 128 -> 128 layer on split-fp16 matrix operands (the tile arithmetic of k_pfn2_h, csrc/pfn_fused.hip), a
//                wave owns fixed 16-row windows (one tile) of the sorted order: its 16-row tiles go through a per-wave LDS tile, lane =
//                channel walks the rows with a running max that is flushed (one 256-byte store) whenever the voxel changes;
//                only the voxels a window boundary cuts (a few per cent) take integer atomic max, on rows k_ps_place zeroed.
//                (Voxel-aligned tasks were measured first: 51 us -- the crowded voxels next to the sensor, hundreds of points
//                each, went through ONE wave tile by tile and set the launch's duration.)
// The order of the points inside a voxel is the arrival order of the rank atomics (not deterministic); every reduction over
// a run is order independent (exact integer sums, max), and a row's layer outputs depend on the row alone -- the results are
// bit-identical run to run (tests/test_vfe_gpu.py asserts it).
#include "common.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
#define PS_SCALE 2048.0f
#define PS_INV (1.0f / 2048.0f)
#define PS_MFMA(acc, av, bv) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16((av), (bv), acc, 0, 0, 0)
#define PS_FIX 1048576.0  // 2^20 steps per metre (csrc/vfe.hip)
#define PS_SCAN 1024      // counts per scan block
#define PS_TASK_DEFAULT 16  // sorted rows per task window of k_ps_pfn2 (MSSVT_PFN_TASK = 16 | 32 | 64: 37.3 / 45.9 / 45.1 us at 160k points)

__global__ void __launch_bounds__(256) k_ps_rank(const int *voxel, long long P, int *count, int *slot) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int v = voxel[p];
    slot[p] = v >= 0 ? atomicAdd(count + v, 1) : -1;
}

// level 1: exclusive scan of the counts inside blocks of PS_SCAN, block totals out
__global__ void __launch_bounds__(256) k_ps_scan1(const int *count, int n, int *local, int *block_sum) {
    __shared__ int wsum[4];
    const int base = blockIdx.x * PS_SCAN + threadIdx.x * 4;
    int c[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k] = base + k < n ? count[base + k] : 0;
        tot += c[k];
    }
    int incl = tot;
    const int lane = lane_id(), wv = threadIdx.x / MSSVT_WAVE;
    for (int off = 1; off < MSSVT_WAVE; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == MSSVT_WAVE - 1) wsum[wv] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wv; ++w) before += wsum[w];
    int run = before + incl - tot;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) local[base + k] = run;
        run += c[k];
    }
    if (threadIdx.x == 255) block_sum[blockIdx.x] = before + incl;
}

// level 2 (one workgroup): exclusive scan of the block totals in place, grand total -> *total
__global__ void __launch_bounds__(1024) k_ps_scan2(int *block_sum, int nblocks, int *total) {
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = lane_id(), wv = threadIdx.x / MSSVT_WAVE;
    for (int base = 0; base < nblocks; base += 1024) {
        const int idx = base + threadIdx.x;
        const int v = idx < nblocks ? block_sum[idx] : 0;
        int incl = v;
        for (int off = 1; off < MSSVT_WAVE; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == MSSVT_WAVE - 1) wsum[wv] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wv; ++w) before += wsum[w];
        if (idx < nblocks) block_sum[idx] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// thread i: point i -> its place in the sorted order; voxel i (i <= N) -> start[i].  k_ps_pfn2 deals the sorted rows in
// fixed windows of task_rows rows; a voxel whose run is CUT by a window boundary gets its maximum from two or more waves through
// integer atomic max on the (non-negative) float bits -- its output row is zeroed here, ahead of them (a few per cent of the voxels)
__global__ void __launch_bounds__(256)
    k_ps_place(const int *voxel, long long P, const int *slot, const int *local, const int *block_sum, const int *total, int N,
               int task_rows, const float *points, int stride, float4 *pts8, int *row_voxel, int *start, float *out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < P) {
        const int v = voxel[i];
        if (v >= 0) {
            const int r = local[v] + block_sum[v / PS_SCAN] + slot[i];
            // the point's row moves to its place in the sorted order ([x, y, z, f4 | f5, 0, 0, 0]: 32 bytes): the per-voxel
            // loops below then read contiguous rows instead of chasing an index into the scan order
            const float *pr = points + i * stride;
            pts8[2 * (size_t)r] = make_float4(pr[1], pr[2], pr[3], pr[4]);
            pts8[2 * (size_t)r + 1] = make_float4(pr[5], 0.f, 0.f, 0.f);
            row_voxel[r] = v;
        }
    }
    if (i <= N) {
        const int v = (int)i;
        const int s = v < N ? local[v] + block_sum[v / PS_SCAN] : *total;
        start[v] = s;
        if (v < N) {
            const int e = v + 1 < N ? local[v + 1] + block_sum[(v + 1) / PS_SCAN] : *total;
            if (s / task_rows != (e - 1) / task_rows) {
                float4 *o = reinterpret_cast<float4 *>(out + (size_t)v * 128);
#pragma unroll 8
                for (int k = 0; k < 32; ++k) o[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}

// cluster centre of every voxel: exact 64-bit fixed-point sums over its run (order independent), the expression of
// k_vfe_sum_xyz / k_vfe_mean_xyz (csrc/vfe.hip).  16 lanes per voxel, lane q takes rows s + q, s + q + 16, ... (a crowded voxel
// -- hundreds of points next to the sensor -- would otherwise set the launch's duration from one thread), integer sums added
// across the lanes (exact: any order)
__device__ __forceinline__ long long ps_row_sum_ll(long long v) {  // all-reduce over the 16 lanes of a voxel's group
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const int lo = __shfl_xor((int)(unsigned int)(unsigned long long)v, off), hi = __shfl_xor((int)((unsigned long long)v >> 32), off);
        v += (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
    }
    return v;
}
__global__ void __launch_bounds__(256) k_ps_mean(const float4 *pts8, int N, const int *start, float *mean3) {
    const int q = threadIdx.x & 15;
    const int v = min(blockIdx.x * 16 + (threadIdx.x >> 4), N - 1);  // (clamped: every lane takes part in the lane sums)
    const int s = start[v], e = start[v + 1];
    long long sx = 0, sy = 0, sz = 0;
    for (int i = s + q; i < e; i += 16) {
        const float4 p = pts8[2 * (size_t)i];
        sx += __double2ll_rn((double)p.x * PS_FIX);
        sy += __double2ll_rn((double)p.y * PS_FIX);
        sz += __double2ll_rn((double)p.z * PS_FIX);
    }
    sx = ps_row_sum_ll(sx); sy = ps_row_sum_ll(sy); sz = ps_row_sum_ll(sz);
    if (q < 3 && blockIdx.x * 16 + (threadIdx.x >> 4) < N) {
        const long long t = q == 0 ? sx : q == 1 ? sy : sz;
        mean3[(size_t)v * 3 + q] = (float)((double)t / PS_FIX / (double)(e - s));
    }
}

struct Ps1Args {
    const float4 *pts8;               // sorted point rows [x, y, z, f4 | f5, 0, 0, 0]
    const int *total;                 // rows of the sorted order (points inside the grid)
    const int *row_voxel;             // sorted row -> voxel
    const float *mean3;               // (N, 3)
    const int *coords;                // (N, 4) [b, z, y, x]
    float vs[3], off[3];              // voxel size, voxel_size / 2 + range_min
    const float *W, *b, *bn_w, *bn_b, *bn_mean, *bn_var;  // W (64, 11)
    float eps;
    float *x1;  // (rows, 64) in sorted order
};

// layer 1 of every point, 8 lanes per SORTED row (8 output channels each: the per-row part -- five loads, the 11 inputs -- is
// paid once per 8 channels): the 11 inputs built on the fly (the reference's expressions, multiply and add rounded separately),
// 88 FMAs per lane, BatchNorm (running statistics) + ReLU; x1 rows leave in sorted order -- contiguous runs per voxel for
// k_ps_max1 / k_ps_pfn2
__global__ void __launch_bounds__(256) k_ps_pfn1(Ps1Args a) {
    __shared__ float Wl[11 * 64], bl[64], sl[64], tl[64], gl[64], hl[64];
    // input-major in the LDS: the 8 lanes of a row read channels 8 q + i of ONE input -- 8 different banks (channel-major
    // rows of 12 floats put all of them on one bank)
    for (int e = threadIdx.x; e < 64 * 11; e += 256) Wl[(e % 11) * 64 + e / 11] = a.W[e];
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        bl[c] = a.b[c];
        sl[c] = 1.0f / sqrtf(a.bn_var[c] + a.eps);  // torch's eval batch norm: (z - mean) * invstd * weight + bias
        tl[c] = a.bn_mean[c];
        gl[c] = a.bn_w[c];
        hl[c] = a.bn_b[c];
    }
    __syncthreads();
    const int q = threadIdx.x & 7;  // channels [8 q, 8 q + 8) of the row
    const long long r = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
    if (r >= *a.total) return;
    const int v = a.row_voxel[r];
    const float4 p0 = a.pts8[2 * (size_t)r], p1 = a.pts8[2 * (size_t)r + 1];
    const float x = p0.x, y = p0.y, z = p0.z;
    const int4 c4 = reinterpret_cast<const int4 *>(a.coords)[v];
    float f[11];
    f[0] = x; f[1] = y; f[2] = z; f[3] = p0.w; f[4] = p1.x;
    f[5] = x - a.mean3[(size_t)v * 3 + 0]; f[6] = y - a.mean3[(size_t)v * 3 + 1]; f[7] = z - a.mean3[(size_t)v * 3 + 2];
    f[8] = x - __fadd_rn(__fmul_rn((float)c4.w, a.vs[0]), a.off[0]);  // ref :107-109: coord * voxel_size + offset
    f[9] = y - __fadd_rn(__fmul_rn((float)c4.z, a.vs[1]), a.off[1]);
    f[10] = z - __fadd_rn(__fmul_rn((float)c4.y, a.vs[2]), a.off[2]);
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = 8 * q + i;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) acc = __builtin_fmaf(f[k], Wl[k * 64 + c], acc);
        acc += bl[c];
        acc = (acc - tl[c]) * sl[c] * gl[c] + hl[c];
        o[i] = fmaxf(acc, 0.f);
    }
    float4 *dst = reinterpret_cast<float4 *>(a.x1 + (size_t)r * 64 + 8 * q);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
}

// m1 = channel-wise max over a voxel's run of x1 rows: 16 lanes per voxel, eight rows in flight
__global__ void __launch_bounds__(256) k_ps_max1(const float *x1, int N, const int *start, float *m1) {
    const int q = threadIdx.x & 15;
    const int v = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (v >= N) return;
    const int s = start[v], e = start[v + 1];
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);  // (x1 >= 0 after the ReLU and every voxel has a point)
    for (int r = s; r < e; r += 8) {
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4 *>(x1 + (size_t)min(r + u, e - 1) * 64 + 4 * q);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            m.x = fmaxf(m.x, t[u].x); m.y = fmaxf(m.y, t[u].y); m.z = fmaxf(m.z, t[u].z); m.w = fmaxf(m.w, t[u].w);
        }
    }
    *reinterpret_cast<float4 *>(m1 + (size_t)v * 64 + 4 * q) = m;
}

__device__ __forceinline__ void ps_split8(const float4 v0, const float4 v1, float s, h16x8 &hi, h16x8 &lo) {
    const float x[8] = {v0.x * s, v0.y * s, v0.z * s, v0.w * s, v1.x * s, v1.y * s, v1.z * s, v1.w * s};
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const fp16x2 a = __builtin_amdgcn_cvt_pkrtz(x[i], x[i + 1]);
        const fp16x2 c = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)a[0], -PS_SCALE, x[i] * PS_SCALE),
                                                    __builtin_fmaf((float)a[1], -PS_SCALE, x[i + 1] * PS_SCALE));
        hi[i] = (_Float16)a[0]; hi[i + 1] = (_Float16)a[1];
        lo[i] = (_Float16)c[0]; lo[i + 1] = (_Float16)c[1];
    }
}

struct Ps2Args {
    int N, task_rows;
    const int *total, *start, *row_voxel;
    const float *x1, *m1;  // (rows, 64) sorted, (N, 64)
    const float *W, *b, *bn_w, *bn_b, *bn_mean, *bn_var;  // W (128, 128): columns [0, 64) <-> x1, [64, 128) <-> m1[voxel]
    float eps;
    float *out;  // (N, 128)
};

#define PS2_WAVES 16
#define PS2_TS 68  // floats per row of a wave's LDS tile (16 rows x 64 channels + pad)
__global__ void __launch_bounds__(PS2_WAVES *MSSVT_WAVE, 1) k_ps_pfn2(Ps2Args a) {
    constexpr int K = 128, N = 128, KS = K / 32, NT = N / 16, IMG = KS * NT * 64;
    extern __shared__ float4 lds4[];
    h16x8 *Bh = reinterpret_cast<h16x8 *>(lds4), *Bl = Bh + IMG;
    float *tiles = reinterpret_cast<float *>(Bl + IMG);
    __shared__ float wmax_l[PS2_WAVES];
    __shared__ float4 ep_l[3][N / 4];  // per channel: bias | running mean | invstd
    constexpr int FR = IMG / (PS2_WAVES * MSSVT_WAVE);
    float4 wv0[FR], wv1[FR];
    float wmx = 0.f;
#pragma unroll
    for (int i = 0; i < FR; ++i) {
        const int f = threadIdx.x + i * PS2_WAVES * MSSVT_WAVE;
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
    for (int i = 0; i < PS2_WAVES; ++i) wmx = fmaxf(wmx, wmax_l[i]);
    const int web = __builtin_bit_cast(int, wmx) & 0x7F800000;
    const bool wnorm = web != 0 && web < 0x7F000000;
    const float w_in = wnorm ? __builtin_bit_cast(float, 0x7F000000 - web) : 1.0f;
    const float w_un = wnorm ? __builtin_bit_cast(float, web) : 1.0f;
#pragma unroll
    for (int i = 0; i < FR; ++i) {
        const int f = threadIdx.x + i * PS2_WAVES * MSSVT_WAVE;
        h16x8 h, l;
        ps_split8(wv0[i], wv1[i], w_in, h, l);
        Bh[f] = h;
        Bl[f] = l;
    }
    const int lane = lane_id(), la = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    float *tile = tiles + wv * 16 * PS2_TS;
    __syncthreads();
    const int total = *a.total;
    const int ntasks = (total + a.task_rows - 1) / a.task_rows, step = gridDim.x * PS2_WAVES;
    // the voxels of a task's rows are requested one task ahead: with them in hand the x1 rows, the m1 gather and the two
    // start[] probes of a task all leave together (one round trip in front of the products instead of two)
    int pf_first = 0, pf_last = 0, pf_lane = 0;
#define PS2_PREFETCH(t_)                                                              \
    {                                                                                 \
        const int rs_ = (t_) * a.task_rows, re_ = min(rs_ + a.task_rows, total);      \
        pf_first = a.row_voxel[rs_];                                                  \
        pf_last = a.row_voxel[re_ - 1];                                               \
        pf_lane = a.row_voxel[min(rs_ + la, re_ - 1)];                                \
    }
    int task = blockIdx.x * PS2_WAVES + wv;
    if (task < ntasks) PS2_PREFETCH(task)
    for (; task < ntasks; task += step) {
        // a fixed window of the sorted order (perfect balance, whatever the voxels' sizes); its first / last voxel may be cut
        // by the window: those two take integer atomic max on zeroed rows (k_ps_place), every other voxel a plain store
        const int rs = task * a.task_rows, re = min(rs + a.task_rows, total);
        const int v_first = __builtin_amdgcn_readfirstlane(pf_first), v_last = __builtin_amdgcn_readfirstlane(pf_last);
        const int lane_v0 = pf_lane;
        if (task + step < ntasks) PS2_PREFETCH(task + step)
        // (raw loads: they are first READ at a flush, behind the tile's products -- a readfirstlane here would wait for them
        // in front of the row loads)
        const int sf_raw = a.start[v_first], sl_raw = a.start[v_last + 1];
        int cur[2] = {-1, -1};      // the voxel whose running max a half holds (wave uniform)
        float run[2] = {0.f, 0.f};  // lane = channel 64 h + lane
#define PS2_FLUSH(h_)                                                                                              \
        if (cur[h_] >= 0) {                                                                                        \
            float *dst_ = a.out + (size_t)cur[h_] * N + 64 * (h_) + lane;                                          \
            if ((cur[h_] == v_first && __builtin_amdgcn_readfirstlane(sf_raw) < rs) ||                             \
                (cur[h_] == v_last && __builtin_amdgcn_readfirstlane(sl_raw) > re))                                \
                atomicMax(reinterpret_cast<int *>(dst_), __builtin_bit_cast(int, run[h_])); /* run >= 0: ordered as ints */ \
            else                                                                                                   \
                *dst_ = run[h_];                                                                                   \
        }
        for (int r0 = rs; r0 < re; r0 += 16) {
            const int row = min(r0 + la, re - 1);
            const int vs = r0 == rs ? lane_v0 : a.row_voxel[row];
            const int v = r0 + la < re ? vs : -2;
            // lane (row = la, g) reads its row's k slots 32 P + 8 g ..: P = 0, 1 from x1[row], P = 2, 3 from m1[voxel]
            float4 xr[KS][2];
#pragma unroll
            for (int P = 0; P < KS; ++P) {
                const float *src = P < 2 ? a.x1 + (size_t)row * 64 + 32 * P + 8 * g : a.m1 + (size_t)vs * 64 + 32 * (P - 2) + 8 * g;
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
            for (int P = 0; P < KS; ++P) ps_split8(xr[P][0], xr[P][1], s_in, ah[P], al[P]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int tt = 0; tt < NT / 2; ++tt) {  // (the four tiles of a half as interleaved chains, k step outermost: 41.6 against 38.4 us)
                    const int t = h * (NT / 2) + tt;
                    f32x4 mm = f32x4{0.f, 0.f, 0.f, 0.f}, cr = mm;
#pragma unroll
                    for (int P = 0; P < KS; ++P) {
                        const h16x8 bh = Bh[(P * NT + t) * 64 + lane], bl = Bl[(P * NT + t) * 64 + lane];
                        PS_MFMA(mm, bh, ah[P]);
                        PS_MFMA(cr, bh, al[P]);
                        PS_MFMA(cr, bl, ah[P]);
                    }
                    // lane (row = la, g) holds x2[row][16 t + 4 g + i]
                    const float4 b4 = ep_l[0][4 * t + g], m4 = ep_l[1][4 * t + g], s4 = ep_l[2][4 * t + g];
                    const float4 w4 = *reinterpret_cast<const float4 *>(a.bn_w + 16 * t + 4 * g),
                                 c4 = *reinterpret_cast<const float4 *>(a.bn_b + 16 * t + 4 * g);
                    float r[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(cr[i], PS_INV, mm[i]) * un;
                    const float o0 = fmaxf(((r[0] + b4.x) - m4.x) * s4.x * w4.x + c4.x, 0.f), o1 = fmaxf(((r[1] + b4.y) - m4.y) * s4.y * w4.y + c4.y, 0.f),
                                o2 = fmaxf(((r[2] + b4.z) - m4.z) * s4.z * w4.z + c4.z, 0.f), o3 = fmaxf(((r[3] + b4.w) - m4.w) * s4.w * w4.w + c4.w, 0.f);
                    *reinterpret_cast<float4 *>(tile + la * PS2_TS + 16 * tt + 4 * g) = make_float4(o0, o1, o2, o3);
                }
                wave_lds_sync();
                // lane = channel 64 h + lane walks the 16 rows; the voxel of a row is wave uniform
                float val[16];  // (all reads first: the uniform branches below are basic-block boundaries the reads cannot cross)
#pragma unroll
                for (int r = 0; r < 16; ++r) val[r] = tile[r * PS2_TS + lane];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int vr = __builtin_amdgcn_readlane(v, r);
                    if (vr != cur[h]) {
                        PS2_FLUSH(h)
                        cur[h] = vr;
                        run[h] = val[r];
                    } else {
                        run[h] = fmaxf(run[h], val[r]);
                    }
                }
                wave_lds_sync();
            }
        }
        PS2_FLUSH(0)
        PS2_FLUSH(1)
#undef PS2_FLUSH
    }
#undef PS2_PREFETCH
}

extern "C" long long mssvt_pfn_sorted_workspace_ints(long long num_points, int num_voxels) {
    const long long nb = (num_voxels + PS_SCAN - 1) / PS_SCAN;
    // count | local | block_sum | total | start (N + 1) | slot | row_voxel (P each) | mean3 (3 N floats) | sorted point rows (8 P floats)
    return 2LL * num_voxels + nb + 1 + 16 + (num_voxels + 1) + 2 * num_points + 3LL * num_voxels + 8 + 8 * num_points;
}

// DynamicVFE's cluster-centre mean and two PFN layers in eval mode (ref dynamic_vfe.py:96-131), default configuration:
// 5 point features, cluster and voxel-centre offsets, NUM_FILTERS [64, 128].  points (P, stride >= 6) f32 rows
// [b, x, y, z, f4, f5]; point_voxel (P) int32 (-1: outside the grid) and voxel_coords (N, 4) int32 [b, z, y, x] from
// mssvt_voxelize (every voxel holds at least one point); host_voxel_size3 / host_offset3: HOST float[3]
// (offset = voxel_size / 2 + range_min); layer parameters as the state dict holds them; workspace:
// mssvt_pfn_sorted_workspace_ints(P, N) int32; x1_scratch (P, 64), m1_scratch (N, 64): caller-owned; out (N, 128).
extern "C" int mssvt_pfn_sorted_64_128(const float *points, int point_stride, long long num_points, const int *point_voxel,
                                       int num_voxels, const int *voxel_coords, const float *host_voxel_size3,
                                       const float *host_offset3, const float *W1, const float *b1, const float *bn1_w,
                                       const float *bn1_b, const float *bn1_mean, const float *bn1_var, float bn1_eps,
                                       const float *W2, const float *b2, const float *bn2_w, const float *bn2_b,
                                       const float *bn2_mean, const float *bn2_var, float bn2_eps, int *workspace,
                                       float *x1_scratch, float *m1_scratch, float *out, void *stream_) {
    if (num_points < 0 || num_voxels < 0 || point_stride < 6 || (num_points > 0 && (!points || !point_voxel)) || !voxel_coords ||
        !host_voxel_size3 || !host_offset3 || !W1 || !b1 || !bn1_w || !bn1_b || !bn1_mean || !bn1_var || !W2 || !b2 || !bn2_w || !bn2_b ||
        !bn2_mean || !bn2_var || !workspace || !x1_scratch || !m1_scratch || !out)
        return MSSVT_E_BADARG;
    if (num_voxels == 0 || num_points == 0) return MSSVT_OK;
    if (num_points > 0x7FFFFFF0LL) return MSSVT_E_TOOLARGE;
    hipStream_t stream = (hipStream_t)stream_;
    const int N = num_voxels, nb = (N + PS_SCAN - 1) / PS_SCAN;
    int *count = workspace, *local = count + N, *block_sum = local + N, *total = block_sum + nb + 1;
    int *start = total + 16, *slot = start + N + 1, *row_voxel = slot + num_points;
    float *mean3 = reinterpret_cast<float *>(row_voxel + num_points);
    // (16-byte aligned: the workspace is, and every region before this one is a whole number of ints)
    float4 *pts8 = reinterpret_cast<float4 *>((reinterpret_cast<uintptr_t>(mean3 + 3 * (size_t)N) + 15) & ~(uintptr_t)15);
    hipError_t e = hipMemsetAsync(count, 0, (size_t)N * sizeof(int), stream);
    if (e != hipSuccess) return (int)e;
    k_ps_rank<<<divup(num_points, 256), 256, 0, stream>>>(point_voxel, num_points, count, slot);
    k_ps_scan1<<<nb, 256, 0, stream>>>(count, N, local, block_sum);
    k_ps_scan2<<<1, 1024, 0, stream>>>(block_sum, nb, total);
    static const int task_env = getenv("MSSVT_PFN_TASK") ? atoi(getenv("MSSVT_PFN_TASK")) : 0;
    const int task_rows = task_env == 16 || task_env == 32 || task_env == 64 ? task_env : PS_TASK_DEFAULT;
    const long long nthreads = num_points > N + 1 ? num_points : N + 1;
    k_ps_place<<<divup(nthreads, 256), 256, 0, stream>>>(point_voxel, num_points, slot, local, block_sum, total, N, task_rows, points,
                                                         point_stride, pts8, row_voxel, start, out);
    k_ps_mean<<<divup(N, 16), 256, 0, stream>>>(pts8, N, start, mean3);
    Ps1Args a1;
    a1.pts8 = pts8; a1.total = total; a1.row_voxel = row_voxel; a1.mean3 = mean3;
    a1.coords = voxel_coords;
    for (int k = 0; k < 3; ++k) { a1.vs[k] = host_voxel_size3[k]; a1.off[k] = host_offset3[k]; }
    a1.W = W1; a1.b = b1; a1.bn_w = bn1_w; a1.bn_b = bn1_b; a1.bn_mean = bn1_mean; a1.bn_var = bn1_var; a1.eps = bn1_eps;
    a1.x1 = x1_scratch;
    k_ps_pfn1<<<divup(num_points, 32), 256, 0, stream>>>(a1);
    k_ps_max1<<<divup(N, 16), 256, 0, stream>>>(x1_scratch, N, start, m1_scratch);
    Ps2Args a2;
    a2.N = N; a2.task_rows = task_rows; a2.total = total; a2.start = start; a2.row_voxel = row_voxel;
    a2.x1 = x1_scratch; a2.m1 = m1_scratch;
    a2.W = W2; a2.b = b2; a2.bn_w = bn2_w; a2.bn_b = bn2_b; a2.bn_mean = bn2_mean; a2.bn_var = bn2_var; a2.eps = bn2_eps; a2.out = out;
    const size_t lds = (size_t)128 * 128 * 4 + (size_t)PS2_WAVES * 16 * PS2_TS * 4;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_ps_pfn2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const int grid = min((long long)cus, (long long)divup(num_points / task_rows + 1, PS2_WAVES));
    k_ps_pfn2<<<grid, PS2_WAVES * MSSVT_WAVE, lds, stream>>>(a2);
    return mssvt_launch_status();
}
