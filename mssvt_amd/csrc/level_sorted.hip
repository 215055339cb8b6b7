// level_sorted.hip -- set-up of a resolution level whose voxel list is SORTED by (b, x, y, z).
//
// That is the order DynamicVFE emits (torch.unique over the keys ((b X + x) Y + y) Z + z, ref
// pcdet/models/backbones_3d/vfe/dynamic_vfe.py:83-93,114-118; csrc/voxelize.hip here), i.e. the order every frame of
// the detector arrives in.  For such a list everything the reference builds with hash tables and atomics
// (K1 ms_sparse_attention_gpu.cu:66-97, K2 :117-168, the .item() loops of mssvt_utils.py:35-37) follows from ONE
// occupancy bitmap, with the canonical orders of the oracle (first occurrence in voxel-index order) for free:
//
//   * index of the voxel in cell (x, y, z) inside its sample = voxels in the columns before (x, y)
//     + popcount(column word below z): no voxel hash table, no probing (k_window_plan reads col_vbase);
//   * the first voxel of a window is its smallest (x, y, z): window (wx, wy, wz) is first seen in the first column of
//     its footprint whose z-slab wz is occupied, and windows are numbered in that (column, slab) order -- exactly the
//     first-occurrence order of K2's canonical form (hash_build.hip) without insert-min / flag / rank passes;
//   * samples are contiguous runs: their starts are the positions where b changes (no counting atomics).
//
// Three launches behind one fill: k_level_mark (per voxel: order check, occupancy bits, sample starts),
// k_col_sums (per 1024 columns: voxels and first-seen windows of every partition), k_col_emit (scan + column bases
// + window rows + window hash tables).  A list that is not strictly ascending / in-grid sets ST_UNSORTED in
// level_status[0] and the column kernels leave everything empty (0 windows): the caller falls back to the
// order-agnostic path (mssvt_level_setup).  Integer work only; bit-exact against the oracle by construction.
#include "common.hip.h"

#define LS_MAX_PARTS 4
#define LS_COLS 1024  // columns per workgroup of the column kernels

struct LsPart {
    int wsx, wsy, wsz;  // window size in voxels
    int gx, gy, gz;     // window grid (spatial_shape // window size: ragged border cells belong to no window)
    int max_wins;       // per sample (the reference's num_windows): ranks beyond get no table entry
    int *win_ind;       // (capacity, 4) [b, wz, wy, wx], window order
    slot_t *table;      // (B, H) window key -> rank inside the sample, or nullptr (nobody reads it)
    int *vcount;        // (B) windows per sample
    int *ws;            // header words: [0] status bits, [1] number of windows (all samples)
};
struct LsArgs {
    const int *indices;
    int n, B, X, Y, Z, H;
    unsigned long long *occ;  // (B, X, Y) bit z
    int *vbase;               // (B, X, Y) voxels of the sample in earlier columns
    int *counts;              // (B) v_bs_cnt
    int *start;               // (B + 1) first row of each sample
    int *status;              // level status word
    int nparts, nblk;
    int *sums;  // (B, nblk, 1 + LS_MAX_PARTS)
    LsPart p[LS_MAX_PARTS];
    // optional (pl_part >= 0): the K4 lists of partition pl_part's windows -- pillar windows [1,1,z]: each is a slab of this very
    // column's occupancy word -- written while the window is emitted (what k_window_plan_pillars, compress.hip, computes
    // in a launch of its own: ref gather_one_window_voxels, ms_sparse_attention_gpu.cu:383-433)
    int pl_part, pl_max, pl_n;  // partition, max_num_win1, table entries
    const int *pl_table;        // (pl_n, 3) offsets; x = y = 0 (the caller checked), z used
    int *pl_k_ind, *pl_vstart, *pl_cnt, *pl_base, *pl_pair_win, *pl_pair_vox;
};

__global__ void __launch_bounds__(256) k_level_mark(LsArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int4 v = reinterpret_cast<const int4 *>(a.indices)[i];  // [b, z, y, x]
    const bool valid = v.x >= 0 && v.x < a.B && v.w >= 0 && v.w < a.X && v.z >= 0 && v.z < a.Y && v.y >= 0 && v.y < a.Z;
    bool bad = !valid;
    int bprev = -1;
    if (i > 0) {
        const int4 u = reinterpret_cast<const int4 *>(a.indices)[i - 1];
        bprev = u.x;
        // strictly ascending in (b, x, y, z): also rules out duplicates
        const bool less = u.x != v.x ? u.x < v.x : u.w != v.w ? u.w < v.w : u.z != v.z ? u.z < v.z : u.y < v.y;
        bad = bad || !less;
    }
    if (__ballot(bad) != 0ull) {
        if (bad) atomicOr(a.status, ST_UNSORTED);  // (wave-aggregated by the compiler)
    }
    if (!valid) return;
    atomicOr(a.occ + ((size_t)v.x * a.X + v.w) * a.Y + v.z, 1ull << v.y);
    // sample starts: row i opens samples (bprev, b]
    if (v.x != bprev) {
        const int lo = bprev < -1 ? -1 : bprev >= a.B ? a.B - 1 : bprev;
        for (int b = lo + 1; b <= v.x; ++b) a.start[b] = i;
    }
    if (i == a.n - 1)
        for (int b = v.x + 1; b <= a.B; ++b) a.start[b] = a.n;
}

// slabs of column (x, y) in which it is the FIRST occupied column of its window's footprint (bit s = slab s)
__device__ __forceinline__ unsigned long long ls_first_slabs(const LsPart &P, const unsigned long long *occ_b, int Y, int x, int y,
                                                            unsigned long long word) {
    if (word == 0ull || x >= P.gx * P.wsx || y >= P.gy * P.wsy) return 0ull;
    const int x0 = x - x % P.wsx, y0 = y - y % P.wsy;
    unsigned long long earlier = 0ull;
    // (all <= 15 words of a footprint requested together, predicated and unrolled, instead of these dependent loads: measured
    // slower, 16.2 -> 17.6 us in k_col_emit: most columns of a window are the first or second of their footprint)
    for (int xx = x0; xx <= x; ++xx) {
        const int yend = xx < x ? y0 + P.wsy : y;
        for (int yy = y0; yy < yend; ++yy) earlier |= occ_b[(size_t)xx * Y + yy];
    }
    const unsigned long long sm = P.wsz >= 64 ? ~0ull : (1ull << P.wsz) - 1ull;
    unsigned long long out = 0ull;
    for (int s = 0; s < P.gz; ++s) {
        const unsigned long long m = sm << (s * P.wsz);
        if ((word & m) != 0ull && (earlier & m) == 0ull) out |= 1ull << s;
    }
    return out;
}

__global__ void __launch_bounds__(LS_COLS) k_col_sums(LsArgs a) {
    if (a.status[0] & ST_UNSORTED) return;
    __shared__ int red[LS_COLS / MSSVT_WAVE][1 + LS_MAX_PARTS];
    const int b = blockIdx.y, c = blockIdx.x * LS_COLS + threadIdx.x, ncol = a.X * a.Y;
    const unsigned long long *occ_b = a.occ + (size_t)b * ncol;
    int v[1 + LS_MAX_PARTS] = {0, 0, 0, 0, 0};
    if (c < ncol) {
        const unsigned long long word = occ_b[c];
        v[0] = __popcll(word);
        if (word) {
            const int x = c / a.Y, y = c % a.Y;
#pragma unroll
            for (int t = 0; t < LS_MAX_PARTS; ++t)
                if (t < a.nparts) v[1 + t] = __popcll(ls_first_slabs(a.p[t], occ_b, a.Y, x, y, word));
        }
    }
#pragma unroll
    for (int t = 0; t <= LS_MAX_PARTS; ++t) {
        const int s = wave_sum_i(v[t]);
        if (lane_id() == 0) red[threadIdx.x / MSSVT_WAVE][t] = s;
    }
    __syncthreads();
    if (threadIdx.x <= LS_MAX_PARTS) {
        int s = 0;
        for (int w = 0; w < LS_COLS / MSSVT_WAVE; ++w) s += red[w][threadIdx.x];
        a.sums[((size_t)b * a.nblk + blockIdx.x) * (1 + LS_MAX_PARTS) + threadIdx.x] = s;
    }
}

__global__ void __launch_bounds__(LS_COLS) k_col_emit(LsArgs a) {
    if (a.status[0] & ST_UNSORTED) return;
    constexpr int NW = LS_COLS / MSSVT_WAVE, NQ = 1 + LS_MAX_PARTS;
    __shared__ int red[NW][2 * NQ];
    __shared__ int pre_own[NQ], pre_all[NQ];  // blocks before this one: of this sample / of the earlier samples
    __shared__ int wtot[NW][NQ];
    // pillar lists: table position of every relative z offset (bit dz + 32 of a window-centred word), and per position the
    // mask of the offsets that come EARLIER in the table -- the slot of a set bit is then one popcount, and a window costs
    // one step per voxel it holds (2.3 on average) instead of one per table entry (32)
    __shared__ int pl_q_of[MSSVT_WAVE];
    __shared__ unsigned long long pl_before[MSSVT_WAVE];
    __shared__ unsigned long long pl_all;
    const int b = blockIdx.y, blk = blockIdx.x, ncol = a.X * a.Y, lane = lane_id(), wv = threadIdx.x / MSSVT_WAVE;
    if (a.pl_part >= 0 && threadIdx.x < MSSVT_WAVE) {
        // ONE load per lane (lane q = table entry q) and an OR scan across the lanes -- walking the table entry by entry
        // was up to 2 x 64 dependent L2 round trips in front of every workgroup's first barrier
        const int q = threadIdx.x;
        const int bit = q < a.pl_n ? a.pl_table[q * 3 + 2] + 32 : -1;  // relative offset dz = bit - 32
        const bool ok = (unsigned int)bit < 64u;
        pl_q_of[q] = 0x7FFFFFFF;
        wave_lds_sync();
        if (ok) atomicMin(&pl_q_of[bit], q);  // (should an offset repeat, its first entry counts: the reference's walk)
        wave_lds_sync();
        if (pl_q_of[q] == 0x7FFFFFFF) pl_q_of[q] = -1;
        const unsigned long long mine = ok ? 1ull << bit : 0ull;
        unsigned long long incl = mine;
        for (int off = 1; off < MSSVT_WAVE; off <<= 1) {
            const unsigned long long t = __shfl_up(incl, off);
            if (lane >= off) incl |= t;
        }
        const unsigned long long prev = __shfl_up(incl, 1);
        pl_before[q] = lane >= 1 ? prev : 0ull;  // the offsets that come EARLIER in the table (indexed by table position)
        if (q == MSSVT_WAVE - 1) pl_all = incl;
    }
    // ---- prefixes over the workgroups' sums ----------------------------------------------------------------
    {
        int own[NQ] = {0, 0, 0, 0, 0}, all[NQ] = {0, 0, 0, 0, 0};
        const int upto = b * a.nblk + blk;  // flattened (sample, block) pairs before this workgroup
        for (int k = threadIdx.x; k < upto; k += LS_COLS) {
            const int *s = a.sums + (size_t)k * NQ;
            const bool same = k >= b * a.nblk;
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                const int val = s[t];
                own[t] += same ? val : 0;
                all[t] += same ? 0 : val;
            }
        }
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            const int so = wave_sum_i(own[t]), sa = wave_sum_i(all[t]);
            if (lane == 0) {
                red[wv][t] = so;
                red[wv][NQ + t] = sa;
            }
        }
        __syncthreads();
        if (threadIdx.x < NQ) {
            int so = 0, sa = 0;
            for (int w = 0; w < NW; ++w) {
                so += red[w][threadIdx.x];
                sa += red[w][NQ + threadIdx.x];
            }
            pre_own[threadIdx.x] = so;
            pre_all[threadIdx.x] = sa;
        }
    }
    // ---- this workgroup's columns ---------------------------------------------------------------------------
    const int c = blk * LS_COLS + threadIdx.x;
    const unsigned long long *occ_b = a.occ + (size_t)b * ncol;
    unsigned long long word = 0ull, first[LS_MAX_PARTS] = {0ull, 0ull, 0ull, 0ull};
    int x = 0, y = 0;
    if (c < ncol) {
        word = occ_b[c];
        x = c / a.Y;
        y = c % a.Y;
        if (word) {
#pragma unroll
            for (int t = 0; t < LS_MAX_PARTS; ++t)
                if (t < a.nparts) first[t] = ls_first_slabs(a.p[t], occ_b, a.Y, x, y, word);
        }
    }
    int v[NQ], ex[NQ];
    v[0] = __popcll(word);
#pragma unroll
    for (int t = 0; t < LS_MAX_PARTS; ++t) v[1 + t] = __popcll(first[t]);
#pragma unroll
    for (int t = 0; t < NQ; ++t) {  // exclusive scan inside the wave
        int incl = v[t];
        for (int off = 1; off < MSSVT_WAVE; off <<= 1) {
            const int u = __shfl_up(incl, off);
            if (lane >= off) incl += u;
        }
        ex[t] = incl - v[t];
        if (lane == MSSVT_WAVE - 1) wtot[wv][t] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        int before = pre_own[t];
        for (int w = 0; w < wv; ++w) before += wtot[w][t];
        ex[t] += before;  // rank inside the sample
    }
    if (c < ncol) {
        a.vbase[(size_t)b * ncol + c] = ex[0];
#pragma unroll
        for (int t = 0; t < LS_MAX_PARTS; ++t) {
            if (t >= a.nparts) continue;
            const LsPart &P = a.p[t];
            unsigned long long f = first[t];
            int rank = ex[1 + t];
            const int wx = x / P.wsx, wy = y / P.wsy;
            while (f) {
                const int wz = __ffsll((long long)f) - 1;
                f &= f - 1ull;
                reinterpret_cast<int4 *>(P.win_ind)[pre_all[1 + t] + rank] = make_int4(b, wz, wy, wx);
                if (t == a.pl_part) {  // the window's K4 list: set bits of its slab in table order (offsets around the slab centre)
                    const int W = pre_all[1 + t] + rank, vstart = a.start[b], cz = wz * P.wsz + P.wsz / 2;
                    int *row = a.pl_k_ind + (size_t)W * a.pl_max;  // pre-filled with -1 by the caller
                    // the column word re-centred on the slab: bit dz + 32 <-> cell cz + dz
                    unsigned long long wrel = (cz >= 32 ? word >> (cz - 32) : word << (32 - cz)) & pl_all;
                    const int cnt = __popcll(wrel);
                    const unsigned long long wfull = wrel;
                    while (wrel) {
                        const int dzb = __ffsll((long long)wrel) - 1;
                        wrel &= wrel - 1ull;
                        const int slot = __popcll(wfull & pl_before[pl_q_of[dzb]]);
                        if (slot < a.pl_max) {
                            const int sz = cz + dzb - 32;
                            const int sv = ex[0] + __popcll(word & ((1ull << sz) - 1ull));
                            if (a.pl_k_ind) row[slot] = sv;  // (optional: mssvt_compress_ws takes the windows as runs of rows)
                            a.pl_pair_win[vstart + sv] = W;
                            if (a.pl_pair_vox) a.pl_pair_vox[vstart + sv] = vstart + sv;
                        }
                    }
                    const int nk = cnt < a.pl_max ? cnt : a.pl_max;
                    a.pl_vstart[W] = vstart;
                    a.pl_cnt[W] = nk;
                    if (a.pl_base) a.pl_base[W] = -1;
                }
                if (P.table && rank < P.max_wins) {  // ref :154-161: the reference writes out of bounds beyond max_wins
                    const int st = table_insert_ordered(wx * P.gy * P.gz + wy * P.gz + wz, rank, a.H, P.table + (size_t)b * a.H);
                    if (st & ST_TABLE_OVERFLOW) atomicOr(P.ws + WS_STATUS, ST_TABLE_OVERFLOW);
                }
                ++rank;
            }
        }
    }
    // ---- totals: the last workgroup of the grid has every other one's sums in its prefixes -------------------
    if (b == a.B - 1 && blk == a.nblk - 1) {
        __syncthreads();
        // per sample totals: the sums again (cheap: B x nblk x NQ ints, once)
        for (int pair = wv; pair < a.B * (1 + a.nparts); pair += NW) {
            const int bb = pair / (1 + a.nparts), t = pair % (1 + a.nparts);
            int s = 0;
            for (int k = lane; k < a.nblk; k += MSSVT_WAVE) s += a.sums[((size_t)bb * a.nblk + k) * NQ + t];
            s = wave_sum_i(s);
            if (lane == 0 && t > 0) {
                a.p[t - 1].vcount[bb] = s;
                if (s > a.p[t - 1].max_wins) atomicOr(a.p[t - 1].ws + WS_STATUS, ST_WIN_OVERFLOW);
            }
        }
        if (threadIdx.x < a.nparts) {
            int mine = 0;
            for (int w = 0; w < NW; ++w) mine += wtot[w][1 + threadIdx.x];
            a.p[threadIdx.x].ws[1] = pre_all[1 + threadIdx.x] + pre_own[1 + threadIdx.x] + mine;
        }
        for (int bb = threadIdx.x; bb < a.B; bb += LS_COLS) a.counts[bb] = a.start[bb + 1] - a.start[bb];
    }
}

extern "C" long long mssvt_level_sorted_scratch_ints(int batch_size, int x_max, int y_max) {
    const long long nblk = ((long long)x_max * y_max + LS_COLS - 1) / LS_COLS;
    return (long long)batch_size * nblk * (1 + LS_MAX_PARTS);
}

struct LsPillars {
    int part, max_win1, n_win1;
    const int *table;
    int *k_ind, *vstart, *cnt, *base, *pair_win, *pair_vox;
};

static int level_setup_sorted_impl(int num_voxels, int batch_size, int x_max, int y_max, int z_max, int hash_size,
                                   const int *v_indices, void *zero_region, long long zero_bytes, int *v_bs_cnt,
                                   int *sample_start, unsigned long long *occ_columns, int *column_vbase,
                                   int *level_status, int num_sets, const int *host_win_grid3,
                                   const int *host_win_size3, const int *host_max_num_wins, int *const *host_win_ind,
                                   int *const *host_tables, int *const *host_vcount, int *const *host_ws,
                                   int *scratch, const LsPillars *pl, void *stream_) {
    const bool precleared = zero_bytes < 0;  // the caller cleared the region itself (mssvt_fill_two, with its other fills)
    if (precleared) zero_bytes = -zero_bytes;
    if (!zero_region || zero_bytes <= 0 || !v_bs_cnt || !sample_start || !occ_columns || !column_vbase || !level_status ||
        !scratch || batch_size <= 0 || hash_size <= 0 || num_voxels < 0 || (!v_indices && num_voxels > 0) || x_max <= 0 ||
        y_max <= 0 || z_max <= 0 || num_sets < 0 || num_sets > LS_MAX_PARTS ||
        (num_sets > 0 && (!host_win_grid3 || !host_win_size3 || !host_max_num_wins || !host_win_ind || !host_tables ||
                          !host_vcount || !host_ws)))
        return MSSVT_E_BADARG;
    if (z_max > 64) return MSSVT_E_TOOLARGE;
    LsArgs a;
    a.indices = v_indices;
    a.n = num_voxels; a.B = batch_size; a.X = x_max; a.Y = y_max; a.Z = z_max; a.H = hash_size;
    a.occ = occ_columns; a.vbase = column_vbase; a.counts = v_bs_cnt; a.start = sample_start; a.status = level_status;
    a.nparts = num_sets;
    a.nblk = divup((long long)x_max * y_max, LS_COLS);
    a.sums = scratch;
    const char *z0 = (const char *)zero_region, *z1 = z0 + zero_bytes;
    auto inside = [&](const void *p, size_t bytes) { return (const char *)p >= z0 && (const char *)p + bytes <= z1; };
    bool ok = inside(sample_start, (size_t)(batch_size + 1) * sizeof(int)) && inside(level_status, sizeof(int)) &&
              inside(occ_columns, (size_t)batch_size * x_max * y_max * sizeof(unsigned long long));
    for (int k = 0; k < LS_MAX_PARTS; ++k) {
        LsPart &P = a.p[k];
        if (k >= num_sets) {
            P = LsPart{1, 1, 1, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr};
            continue;
        }
        const int *g = host_win_grid3 + 3 * k, *w = host_win_size3 + 3 * k;
        if (!host_win_ind[k] || !host_vcount[k] || !host_ws[k] || w[0] <= 0 || w[1] <= 0 || w[2] <= 0 || g[0] < 0 ||
            g[1] < 0 || g[2] < 0 || (long long)g[0] * w[0] > x_max || (long long)g[1] * w[1] > y_max ||
            (long long)g[2] * w[2] > z_max)
            return MSSVT_E_BADARG;
        P.wsx = w[0]; P.wsy = w[1]; P.wsz = w[2];
        P.gx = g[0]; P.gy = g[1]; P.gz = g[2];
        P.max_wins = host_max_num_wins[k];
        P.win_ind = host_win_ind[k];
        P.table = reinterpret_cast<slot_t *>(host_tables[k]);
        P.vcount = host_vcount[k];
        P.ws = host_ws[k];
        ok = ok && inside(P.ws, WS_HDR_INTS * sizeof(int));
    }
    if (!ok) return MSSVT_E_BADARG;
    a.pl_part = -1;
    a.pl_max = a.pl_n = 0;
    a.pl_table = nullptr;
    a.pl_k_ind = a.pl_vstart = a.pl_cnt = a.pl_base = a.pl_pair_win = a.pl_pair_vox = nullptr;
    if (pl) {
        if (pl->part < 0 || pl->part >= num_sets || pl->max_win1 <= 0 || pl->n_win1 < 1 || pl->n_win1 > MSSVT_WAVE || !pl->table ||
            !pl->vstart || !pl->cnt || !pl->pair_win)
            return MSSVT_E_BADARG;
        if (a.p[pl->part].wsx != 1 || a.p[pl->part].wsy != 1) return MSSVT_E_TOOLARGE;  // pillar windows only
        a.pl_part = pl->part; a.pl_max = pl->max_win1; a.pl_n = pl->n_win1; a.pl_table = pl->table;
        a.pl_k_ind = pl->k_ind; a.pl_vstart = pl->vstart; a.pl_cnt = pl->cnt; a.pl_base = pl->base;
        a.pl_pair_win = pl->pair_win; a.pl_pair_vox = pl->pair_vox;
    }
    hipStream_t stream = (hipStream_t)stream_;
    if (!precleared) {
        hipError_t e = hipMemsetAsync(zero_region, 0, (size_t)zero_bytes, stream);
        if (e != hipSuccess) return (int)e;
    }
    if (num_voxels > 0) k_level_mark<<<divup(num_voxels, 256), 256, 0, stream>>>(a);
    const dim3 grid(a.nblk, batch_size);
    k_col_sums<<<grid, LS_COLS, 0, stream>>>(a);
    k_col_emit<<<grid, LS_COLS, 0, stream>>>(a);
    return mssvt_launch_status();
}

extern "C" int mssvt_level_setup_sorted(int num_voxels, int batch_size, int x_max, int y_max, int z_max, int hash_size,
                                        const int *v_indices, void *zero_region, long long zero_bytes, int *v_bs_cnt,
                                        int *sample_start, unsigned long long *occ_columns, int *column_vbase,
                                        int *level_status, int num_sets, const int *host_win_grid3,
                                        const int *host_win_size3, const int *host_max_num_wins, int *const *host_win_ind,
                                        int *const *host_tables, int *const *host_vcount, int *const *host_ws,
                                        int *scratch, void *stream) {
    return level_setup_sorted_impl(num_voxels, batch_size, x_max, y_max, z_max, hash_size, v_indices, zero_region, zero_bytes,
                                   v_bs_cnt, sample_start, occ_columns, column_vbase, level_status, num_sets, host_win_grid3,
                                   host_win_size3, host_max_num_wins, host_win_ind, host_tables, host_vcount, host_ws, scratch,
                                   nullptr, stream);
}

extern "C" int mssvt_level_setup_sorted_pillars(
    int num_voxels, int batch_size, int x_max, int y_max, int z_max, int hash_size, const int *v_indices, void *zero_region,
    long long zero_bytes, int *v_bs_cnt, int *sample_start, unsigned long long *occ_columns, int *column_vbase,
    int *level_status, int num_sets, const int *host_win_grid3, const int *host_win_size3, const int *host_max_num_wins,
    int *const *host_win_ind, int *const *host_tables, int *const *host_vcount, int *const *host_ws, int *scratch,
    int pillar_set, int max_num_win1, int num_win1, const int *vox_query_win1, int *k_ind, int *win_vstart, int *win_cnt,
    int *pair_base, int *pair_win, int *pair_vox, void *stream) {
    const LsPillars pl = {pillar_set, max_num_win1, num_win1, vox_query_win1, k_ind, win_vstart, win_cnt, pair_base, pair_win, pair_vox};
    return level_setup_sorted_impl(num_voxels, batch_size, x_max, y_max, z_max, hash_size, v_indices, zero_region, zero_bytes,
                                   v_bs_cnt, sample_start, occ_columns, column_vbase, level_status, num_sets, host_win_grid3,
                                   host_win_size3, host_max_num_wins, host_win_ind, host_tables, host_vcount, host_ws, scratch,
                                   &pl, stream);
}

// ---- two constant fills in one launch: the frame's -1 arena (tables, owner arrays, list prefills) and its zero region
// (status words, window headers, sample starts, occupancy words) were two framework / runtime fill launches
__global__ void __launch_bounds__(256) k_fill_two(int *a, long long n_a, int value_a, int *b, long long n_b, int value_b) {
    const long long q_a = n_a >> 2, q_b = n_b >> 2, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < q_a + q_b; i += stride) {
        if (i < q_a) reinterpret_cast<int4 *>(a)[i] = make_int4(value_a, value_a, value_a, value_a);
        else reinterpret_cast<int4 *>(b)[i - q_a] = make_int4(value_b, value_b, value_b, value_b);
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) {  // the (at most 3 + 3) ints behind the last full quad of each region
        const int t = threadIdx.x & 3;
        if (threadIdx.x < 4) { if ((q_a << 2) + t < n_a) a[(q_a << 2) + t] = value_a; }
        else if ((q_b << 2) + t < n_b) b[(q_b << 2) + t] = value_b;
    }
}

extern "C" int mssvt_fill_two(int *a, long long n_a, int value_a, int *b, long long n_b, int value_b, void *stream) {
    if (n_a < 0 || n_b < 0 || (n_a > 0 && !a) || (n_b > 0 && !b)) return MSSVT_E_BADARG;
    if ((n_a > 0 && ((uintptr_t)a & 15)) || (n_b > 0 && ((uintptr_t)b & 15))) return MSSVT_E_BADARG;  // 16-byte stores
    if (n_a + n_b == 0) return MSSVT_OK;
    long long quads = (n_a >> 2) + (n_b >> 2);
    int grid = (int)((quads + 255) / 256 > 4096 ? 4096 : (quads + 255) / 256);
    if (grid < 1) grid = 1;
    k_fill_two<<<grid, 256, 0, (hipStream_t)stream>>>(a, n_a, value_a, b, n_b, value_b);
    return mssvt_launch_status();
}
