// dense_bev.hip -- sparse voxel set -> dense channel-first grid, the step right after the path.
//
// Replaces SparseTensor.dense() + the view of HeightCompression (ref: scatter_nd / dense,
// pcdet/models/model_utils/mssvt_utils.py:6-19,50-62; height_compression.py:41-45):
//     dense = zeros(B, Z, Y, X, C); dense[b, z, y, x] = features; out = dense.permute(0, 4, 1, 2, 3).contiguous()
//     spatial_features = out.view(B, C * Z, Y, X)
// i.e. a zero fill of the whole grid, a scatter and a 5-D permute copy (three passes over B*C*Z*Y*X floats).
// Here the output is produced in ONE pass as a GATHER: one wavefront owns 64 consecutive x cells of a
// (b, z, y) line, finds each cell's voxel through the set's hash table (key -> row; the table of a
// CompressBlock output is its window table), moves the occupied cells' rows through a small LDS tile and
// writes every channel as one coalesced 256-byte line -- zeros included, so there is no separate fill and
// no write is ever narrower than a full line.  Pure copy: bit-exact.
#include "common.hip.h"

#define DB_WPB 4
#define DB_CH 32  // channels per LDS tile

__global__ void __launch_bounds__(DB_WPB *MSSVT_WAVE)
    k_dense_bev(const float *features, const slot_t *table, const int *v_bs_cnt, int B, int X, int Y, int Z, int C,
                int hash_size, float *out) {
    __shared__ float tile[DB_WPB][DB_CH][MSSVT_WAVE + 1];
    const int wv = threadIdx.x / MSSVT_WAVE, lane = lane_id();
    const int xblocks = (X + MSSVT_WAVE - 1) / MSSVT_WAVE, cgroups = (C + DB_CH - 1) / DB_CH;
    const long long lines = (long long)B * Z * Y * xblocks, items = lines * cgroups;
    // one work item = (64 cells of a line, 32 channels): the gather of a line's occupied cells is a chain of dependent loads
    // (a cell pair per step), so the channel groups of a line go to DIFFERENT waves -- 4 x the waves in flight at C = 128, each
    // with a quarter of the chain (68 -> 56.6 us at the detector's BEV grid: the 113 MB of line stores now set the time); the hash
    // probe is repeated per group (cache hits)
    for (long long item = (long long)blockIdx.x * DB_WPB + wv; item < items; item += (long long)gridDim.x * DB_WPB) {
        const int c0 = (int)(item % cgroups) * DB_CH;
        const long long line = item / cgroups;
        const int xb = (int)(line % xblocks);
        const int y = (int)((line / xblocks) % Y);
        const int z = (int)((line / ((long long)xblocks * Y)) % Z);
        const int b = (int)(line / ((long long)xblocks * Y * Z));
        const int x = xb * MSSVT_WAVE + lane;
        int vstart = 0;
        for (int k = 0; k < b; ++k) vstart += v_bs_cnt[k];
        int row = -1;
        if (x < X) {
            const int sv = table_find(x * Y * Z + y * Z + z, hash_size, table + (size_t)b * hash_size);  // x-major key
            if (sv != MSSVT_EMPTY) row = vstart + sv;
        }
        const unsigned long long occ = __ballot(row >= 0);
        float *dst = out + ((size_t)b * C * Z + z) * Y * X + (size_t)y * X + x;  // + c * Z*Y*X
        const size_t cstride = (size_t)Z * Y * X;
        // occupied cells' rows -> tile[channel][cell]: two cells per step (32 channels each, 128-byte reads)
        unsigned long long m = occ;
        while (m) {
            const int l0 = __ffsll((long long)m) - 1;
            m &= m - 1;
            int l1 = l0;
            if (m) {
                l1 = __ffsll((long long)m) - 1;
                m &= m - 1;
            }
            const int src_lane = lane < DB_CH ? l0 : l1;
            const int r = __shfl(row, src_lane);
            const int c = c0 + (lane & (DB_CH - 1));
            if (c < C && (lane < DB_CH || l1 != l0)) tile[wv][lane & (DB_CH - 1)][src_lane] = features[(size_t)r * C + c];
        }
        wave_lds_sync();
        if (x < X) {
#pragma unroll 8
            for (int cc = 0; cc < DB_CH; ++cc) {
                if (c0 + cc >= C) break;
                dst[(size_t)(c0 + cc) * cstride] = row >= 0 ? tile[wv][cc][lane] : 0.0f;
            }
        }
        wave_lds_sync();
    }
}

// Round 6: the same gather with 16-BYTE stores and no LDS.  A (b, c) plane of the output is Z*Y*X contiguous floats; when
// that count is a multiple of 4 every plane starts 16-byte aligned and a lane owns FOUR consecutive cells of it.  One wave =
// 256 consecutive cells x DB4_CH channels: the four hash probes per lane once, then per chunk of DB4_CK = 8 channels the feature
// values of the occupied cells (two exec-masked 16-byte loads per cell: ~1/6 of the cells of a BEV grid are occupied)
// and one global_store_dwordx4 per channel -- zeros included, so no fill pass and
// every store instruction writes 1 KB of one plane.  The next chunk's loads are issued before this chunk's stores (vmcnt
// retires in order).  56.6 -> see DESIGN (113 MB grid: the write stream is the floor, 113 MB / 6.3 TB/s = 18 us).
#ifndef DB4_CH
#define DB4_CH 32
#endif
#define DB4_CK 8
__global__ void __launch_bounds__(256)
    k_dense_bev4(const float *features, const slot_t *table, const int *v_bs_cnt, int B, int X, int Y, int Z, int C,
                 int hash_size, float *out) {
    const int lane = lane_id(), wv = threadIdx.x / MSSVT_WAVE;
    const long long plane = (long long)Z * Y * X;
    const int blocks = (int)((plane / 4 + MSSVT_WAVE - 1) / MSSVT_WAVE), cgroups = (C + DB4_CH - 1) / DB4_CH;
    const long long items = (long long)B * blocks * cgroups;
    for (long long item = (long long)blockIdx.x * 4 + wv; item < items; item += (long long)gridDim.x * 4) {
        // channel group fastest: the waves of a workgroup share their cells' rows (L1) and probes (L2)
        const int c0 = (int)(item % cgroups) * DB4_CH;
        const int blk = (int)((item / cgroups) % blocks);
        const int b = (int)(item / ((long long)cgroups * blocks));
        const long long cell0 = ((long long)blk * MSSVT_WAVE + lane) * 4;
        int vstart = 0;
        for (int k = 0; k < b; ++k) vstart += v_bs_cnt[k];
        const slot_t *tab = table + (size_t)b * hash_size;
        // the four probes of a lane: their FIRST slots leave together (one round trip; a table at load 0.1 answers nearly every
        // probe there), only a collision walks on through table_find
        int row[4], key[4];
        slot_t first[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long cell = cell0 + j;
            key[j] = -1;
            first[j] = SLOT_EMPTY;
            if (cell < plane) {
                const int x = (int)(cell % X), y = (int)((cell / X) % Y), z = (int)(cell / ((long long)X * Y));
                key[j] = x * Y * Z + y * Z + z;  // x-major key
                first[j] = tab[key[j] % hash_size];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            row[j] = -1;
            if (key[j] >= 0) {
                const int k0 = slot_key(first[j]);
                int sv = MSSVT_EMPTY;
                if (k0 == key[j]) sv = slot_val(first[j]);
                else if (k0 != MSSVT_EMPTY) sv = table_find(key[j], hash_size, tab);
                if (sv != MSSVT_EMPTY) row[j] = vstart + sv;
            }
        }
        if (cell0 >= plane) continue;
        float *dst = out + ((size_t)b * C + c0) * plane + cell0;
        const int nch = min(DB4_CH, C - c0);
        // a cell's DB4_CK = 8 channels of a chunk are two 16-byte loads (C % 8 == 0: aligned); one exec mask per cell
        float4 v[2][4][2];
#define DB4_LOAD(buf_, ck_)                                                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
            v[buf_][j][0] = v[buf_][j][1] = make_float4(0.f, 0.f, 0.f, 0.f);                             \
            if (row[j] >= 0 && (ck_) < nch) {                                                            \
                const float4 *src_ = reinterpret_cast<const float4 *>(features + (size_t)row[j] * C + c0 + (ck_)); \
                v[buf_][j][0] = src_[0];                                                                 \
                v[buf_][j][1] = src_[1];                                                                 \
            }                                                                                            \
        }
#define DB4_PUT(cc_, comp_, h_)                                                                          \
        *reinterpret_cast<float4 *>(dst + (size_t)(ck + (cc_)) * plane) =                                \
            make_float4(v[cur][0][h_].comp_, v[cur][1][h_].comp_, v[cur][2][h_].comp_, v[cur][3][h_].comp_);
        DB4_LOAD(0, 0)
#pragma unroll
        for (int ck = 0; ck < DB4_CH; ck += DB4_CK) {
            const int cur = (ck / DB4_CK) & 1;
            if (ck + DB4_CK < DB4_CH) {
                if (cur == 0) { DB4_LOAD(1, ck + DB4_CK) } else { DB4_LOAD(0, ck + DB4_CK) }
            }
            if (ck < nch) {
                DB4_PUT(0, x, 0) DB4_PUT(1, y, 0) DB4_PUT(2, z, 0) DB4_PUT(3, w, 0)
                DB4_PUT(4, x, 1) DB4_PUT(5, y, 1) DB4_PUT(6, z, 1) DB4_PUT(7, w, 1)
            }
        }
#undef DB4_PUT
#undef DB4_LOAD
    }
}

extern "C" int mssvt_dense_bev(const float *features, int C, const int *map_table, int hash_size,
                               const int *v_bs_cnt, int batch_size, int x_max, int y_max, int z_max, float *out,
                               void *stream) {
    if (!features || !map_table || !v_bs_cnt || !out || C <= 0 || hash_size <= 0 || batch_size <= 0 || x_max <= 0 ||
        y_max <= 0 || z_max <= 0)
        return MSSVT_E_BADARG;
    const long long plane = (long long)z_max * y_max * x_max;
    if (plane % 4 == 0 && C % DB4_CK == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(features) & 15) == 0 && plane / 4 < (1ll << 30)) {
        const long long blocks = (plane / 4 + MSSVT_WAVE - 1) / MSSVT_WAVE;
        const long long items = (long long)batch_size * blocks * ((C + DB4_CH - 1) / DB4_CH);
        long long grid = (items + 3) / 4;
        if (grid > 65536) grid = 65536;
        k_dense_bev4<<<(int)grid, 256, 0, (hipStream_t)stream>>>(
            features, reinterpret_cast<const slot_t *>(map_table), v_bs_cnt, batch_size, x_max, y_max, z_max, C, hash_size, out);
        return mssvt_launch_status();
    }
    const long long lines = (long long)batch_size * z_max * y_max * ((x_max + MSSVT_WAVE - 1) / MSSVT_WAVE) * ((C + DB_CH - 1) / DB_CH);
    long long grid = (lines + DB_WPB - 1) / DB_WPB;
    if (grid > 65536) grid = 65536;
    k_dense_bev<<<(int)grid, DB_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(
        features, reinterpret_cast<const slot_t *>(map_table), v_bs_cnt, batch_size, x_max, y_max, z_max, C, hash_size, out);
    return mssvt_launch_status();
}
