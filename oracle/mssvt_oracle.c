/*
 * mssvt_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the CUDA kernels on MsSVT's mixed-scale
 * sparse-voxel attention hot path.  The loops
 * over windows / rows (one CUDA thread or block each in the reference, no
 * interaction between iterations) carry an OpenMP `parallel for`; the results
 * do not depend on the thread count (tests/test_oracle_golden.py asserts it).
 * K1 / K2 stay sequential: their insertion ORDER is the canonical order.  The
 * gradient kernels (K6, K11) stay sequential too: their sum order is fixed.
 * orc_set_num_threads(1) gives the scalar port bench.py also reports.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (mssvt_amd/) never does.
 *
 * The reference kernels are CUDA-only and cannot be compiled or run here
 * (no nvcc / no GPU in the build container), and the
 * reference ships no tests or golden vectors (F2).  PARITY STATUS:
 *   - kernel level (this file): "parity unpinned" against real CUDA output;
 *     each function follows the cited .cu statement for statement and is
 *     checked against hand-derived known-answer cases in tests/.
 *   - Python level (attention math, query tables, Block / CompressBlock /
 *     backbone orchestration): pinned -- the reference's own Python is
 *     imported in the build container with this library underneath and its
 *     outputs are committed as the tests/golden/ npz fixtures (oracle/gen_golden.py).
 *
 * The reference CUDA path is non-deterministic in three places;
 * "bit-exact" is defined against these canonical orders, each a legal outcome
 * of the CUDA code:
 *   (a) hash-slot layout  = sequential insertion in voxel-index order;
 *   (b) window numbering  = first occurrence in voxel-index order per sample;
 *   (c) duplicate keys    = the last writer (highest voxel index) owns the value.
 *
 * Reference citations use paths relative to /root/reference/pcdet/ops/.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* threads used by the parallel loops below (0 / negative: the OpenMP default) */
static int g_threads = 0;
void orc_set_num_threads(int n) { g_threads = n; }
int orc_get_max_threads(void) {
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}
#define ORC_THREADS (orc_get_max_threads())

#define EMPTY_KEY (-1) /* mssvt/src/ms_cuda_utils.h:9 */

/* ---- hash primitives: mssvt/src/ms_sparse_attention_gpu.cu:18-64 -------- */

/* hash(): :18-20.  key % hash_size (the murmur variant at :8-16 is dead code). */
static int hash_slot(int key, int hash_size) { return key % hash_size; }

/* hash_table_insert(): :22-41, executed sequentially (canonical order (a)).
 * atomicCAS(slot.key, EMPTY, key) becomes a plain compare-and-store.        */
static void table_insert(int key, int value, int hash_size, int *tab) {
    int h = hash_slot(key, hash_size);
    int prob_cnt = 0;
    for (;;) {
        int prev = tab[h * 2 + 0];
        if (prev == EMPTY_KEY) tab[h * 2 + 0] = key;
        if (prev == EMPTY_KEY || prev == key) {
            tab[h * 2 + 1] = value;
            break;
        }
        h = (h + 1) % hash_size;
        prob_cnt += 1;
        if (prob_cnt >= hash_size) break; /* table full: silent drop (:39) */
    }
}

/* hash_table_find(): :43-64 */
static int table_find(int key, int hash_size, const int *tab) {
    int h = key % hash_size;
    int v = EMPTY_KEY;
    int prob_cnt = 0;
    for (;;) {
        if (tab[h * 2 + 0] == key) {
            v = tab[h * 2 + 1];
            break;
        }
        if (tab[h * 2 + 0] == EMPTY_KEY) break;
        h = (h + 1) % hash_size;
        prob_cnt += 1;
        if (prob_cnt >= hash_size) break;
    }
    return v;
}

/* K1  build_mapping_with_hash_kernel: :66-97.
 * v_indices (N,4) [b,z,y,x]; v_bs_cnt (B); table (B,H,2) pre-filled with -1
 * by the caller (mssvt/mssvt_ops.py:16-17).                                 */
int orc_build_mapping_with_hash(int x_max, int y_max, int z_max, int num_voxels,
                                int hash_size, const int *v_indices,
                                const int *v_bs_cnt, int *table) {
    for (int t = 0; t < num_voxels; ++t) {
        int b = v_indices[t * 4 + 0];
        int z = v_indices[t * 4 + 1];
        int y = v_indices[t * 4 + 2];
        int x = v_indices[t * 4 + 3];
        int v_sum = 0;
        for (int k = b - 1; k >= 0; --k) v_sum += v_bs_cnt[k]; /* :81-86 */
        int v_idx = t - v_sum;
        if (x >= x_max || x < 0 || y < 0 || y >= y_max || z < 0 || z >= z_max)
            continue; /* :90 */
        int key = x * y_max * z_max + y * z_max + z; /* :93, x-major */
        table_insert(key, v_idx, hash_size, table + (size_t)b * hash_size * 2);
    }
    return 0;
}

/* K2  window_with_hash_kernel: :117-168.
 * w_indices (B,num_windows,3) pre-filled -1, table (B,H,2) pre-filled -1,
 * vcount (B) zeroed (mssvt/mssvt_ops.py:36-41).  Sequential execution gives
 * canonical order (b).  The reference does not bound-check vcount against
 * num_windows; the oracle returns -1 instead of writing OOB.    */
int orc_window_with_hash(int x_wgs, int y_wgs, int z_wgs, int x_ws, int y_ws,
                         int z_ws, int num_voxels, int num_windows,
                         int hash_size, const int *v_indices, int *w_indices,
                         int *table, int *vcount) {
    for (int t = 0; t < num_voxels; ++t) {
        int b = v_indices[t * 4 + 0];
        int z = v_indices[t * 4 + 1];
        int y = v_indices[t * 4 + 2];
        int x = v_indices[t * 4 + 3];
        int wz = z / z_ws, wy = y / y_ws, wx = x / x_ws; /* C division, :137-139 */
        if (wx < 0 || wx >= x_wgs || wy < 0 || wy >= y_wgs || wz < 0 || wz >= z_wgs)
            continue; /* :141 */
        int *tab = table + (size_t)b * hash_size * 2;
        int *w = w_indices + (size_t)b * num_windows * 3;
        int key = wx * y_wgs * z_wgs + wy * z_wgs + wz; /* :146 */
        int h = hash_slot(key, hash_size);
        int prob_cnt = 0;
        for (;;) {
            int prev = tab[h * 2 + 0];
            if (prev == EMPTY_KEY) {
                tab[h * 2 + 0] = key;
                int v = vcount[b]++;
                if (v >= num_windows) return -1;
                w[v * 3 + 0] = wz;
                w[v * 3 + 1] = wy;
                w[v * 3 + 2] = wx;
                tab[h * 2 + 1] = v;
                break;
            } else if (prev == key) {
                break;
            }
            h = (h + 1) % hash_size;
            prob_cnt += 1;
            if (prob_cnt >= hash_size) break;
        }
    }
    return 0;
}

/* helper for K3/K4: one offset table walked in order, feeding up to three
 * (ind, coord) lists; mirrors the four loops at :227-347.                  */
typedef struct {
    int *ind;   /* row of the ind array for this window   */
    int *coord; /* row of the coord array for this window */
    int max_num;
    int cnt;
} hit_list_t;

static int walk_table(const int *query, int num_query, int cx, int cy, int cz,
                      int x_max, int y_max, int z_max, int hash_size,
                      const int *tab, hit_list_t **app, int n_app,
                      hit_list_t **chk, int n_chk) {
    for (int q = 0; q < num_query; ++q) {
        int ox = query[q * 3 + 0], oy = query[q * 3 + 1], oz = query[q * 3 + 2];
        int sx = cx + ox, sy = cy + oy, sz = cz + oz;
        if (sx >= x_max || sx < 0 || sy >= y_max || sy < 0 || sz >= z_max || sz < 0)
            continue;
        int skey = sx * y_max * z_max + sy * z_max + sz;
        int sv = table_find(skey, hash_size, tab);
        if (sv != EMPTY_KEY) {
            if (n_chk > 0) { /* early return when every listed list is full */
                int all_full = 1;
                for (int l = 0; l < n_chk; ++l)
                    if (chk[l]->cnt < chk[l]->max_num) all_full = 0;
                if (all_full) return 1;
            }
            for (int l = 0; l < n_app; ++l) {
                hit_list_t *L = app[l];
                if (L->cnt < L->max_num) {
                    L->ind[L->cnt] = sv;
                    L->coord[L->cnt * 3 + 0] = ox;
                    L->coord[L->cnt * 3 + 1] = oy;
                    L->coord[L->cnt * 3 + 2] = oz;
                    L->cnt++;
                }
            }
        }
    }
    return 0;
}

/* K3  gather_two_window_voxels_with_hash_kernel: :193-350.
 * Outputs pre-filled by the caller: ind = -1, coord = 0 (mssvt_ops.py:77-85). */
int orc_gather_two_window_voxels(
    int x_max, int y_max, int z_max, int x_ws, int y_ws, int z_ws,
    int max_num_odd, int max_num_even, int max_num_win1, int max_num_win2,
    int num_wins, int hash_size, int num_odd, int num_even, int num_win1,
    int num_win2, int *vox_ind_odd, int *vox_ind_even, int *vox_ind_win1,
    int *vox_ind_win2, int *vox_coord_odd, int *vox_coord_even,
    int *vox_coord_win1, int *vox_coord_win2, const int *vox_query_odd,
    const int *vox_query_even, const int *vox_query_win1,
    const int *vox_query_win2, const int *win_indices, const int *table) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(ORC_THREADS)
    for (int t = 0; t < num_wins; ++t) {
        int b = win_indices[t * 4 + 0];
        int wz = win_indices[t * 4 + 1];
        int wy = win_indices[t * 4 + 2];
        int wx = win_indices[t * 4 + 3];
        const int *tab = table + (size_t)b * hash_size * 2;
        int cx = wx * x_ws + x_ws / 2; /* :219-225 */
        int cy = wy * y_ws + y_ws / 2;
        int cz = wz * z_ws + z_ws / 2;
        hit_list_t odd = {vox_ind_odd + (size_t)t * max_num_odd,
                          vox_coord_odd + (size_t)t * max_num_odd * 3, max_num_odd, 0};
        hit_list_t even = {vox_ind_even + (size_t)t * max_num_even,
                           vox_coord_even + (size_t)t * max_num_even * 3, max_num_even, 0};
        hit_list_t win1 = {vox_ind_win1 + (size_t)t * max_num_win1,
                           vox_coord_win1 + (size_t)t * max_num_win1 * 3, max_num_win1, 0};
        hit_list_t win2 = {vox_ind_win2 + (size_t)t * max_num_win2,
                           vox_coord_win2 + (size_t)t * max_num_win2 * 3, max_num_win2, 0};
        { /* odd hits -> {odd, win1, win2}; the full-check at :238 spans all 4 */
            hit_list_t *app[3] = {&odd, &win1, &win2};
            hit_list_t *chk[4] = {&odd, &even, &win1, &win2};
            if (walk_table(vox_query_odd, num_odd, cx, cy, cz, x_max, y_max, z_max,
                           hash_size, tab, app, 3, chk, 4))
                continue;
        }
        { /* even hits -> {even, win1, win2}; full-check :274 */
            hit_list_t *app[3] = {&even, &win1, &win2};
            if (walk_table(vox_query_even, num_even, cx, cy, cz, x_max, y_max, z_max,
                           hash_size, tab, app, 3, app, 3))
                continue;
        }
        { /* win1_other hits -> {win1, win2}; full-check :310 */
            hit_list_t *app[2] = {&win1, &win2};
            if (walk_table(vox_query_win1, num_win1, cx, cy, cz, x_max, y_max, z_max,
                           hash_size, tab, app, 2, app, 2))
                continue;
        }
        { /* win2_other hits -> {win2}; no early return (:338-346) */
            hit_list_t *app[1] = {&win2};
            walk_table(vox_query_win2, num_win2, cx, cy, cz, x_max, y_max, z_max,
                       hash_size, tab, app, 1, 0, 0);
        }
    }
    return 0;
}

/* K4  gather_one_window_voxels_with_hash_kernel: :383-433 */
int orc_gather_one_window_voxels(int x_max, int y_max, int z_max, int x_ws,
                                 int y_ws, int z_ws, int max_num_win1,
                                 int num_wins, int hash_size, int num_win1,
                                 int *vox_ind_win1, int *vox_coord_win1,
                                 const int *vox_query_win1,
                                 const int *win_indices, const int *table) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(ORC_THREADS)
    for (int t = 0; t < num_wins; ++t) {
        int b = win_indices[t * 4 + 0];
        int wz = win_indices[t * 4 + 1];
        int wy = win_indices[t * 4 + 2];
        int wx = win_indices[t * 4 + 3];
        const int *tab = table + (size_t)b * hash_size * 2;
        int cx = wx * x_ws + x_ws / 2;
        int cy = wy * y_ws + y_ws / 2;
        int cz = wz * z_ws + z_ws / 2;
        hit_list_t win1 = {vox_ind_win1 + (size_t)t * max_num_win1,
                           vox_coord_win1 + (size_t)t * max_num_win1 * 3, max_num_win1, 0};
        hit_list_t *app[1] = {&win1};
        walk_table(vox_query_win1, num_win1, cx, cy, cz, x_max, y_max, z_max,
                   hash_size, tab, app, 1, 0, 0);
    }
    return 0;
}

/* K5  group_features_kernel_stack: mssvt/src/group_features_gpu.cu:73-106.
 * features (N,C), idx (M,nsample), out (M,C,nsample) pre-zeroed by caller.  */
int orc_group_features(int B, int M, int C, int nsample, const float *features,
                       const int *features_batch_cnt, const int *idx,
                       const int *idx_batch_cnt, float *out) {
#pragma omp parallel for schedule(static) num_threads(ORC_THREADS)
    for (int pt = 0; pt < M; ++pt) {
        int bs_idx = 0, pt_cnt = idx_batch_cnt[0]; /* :91-96 */
        for (int k = 1; k < B; k++) {
            if (pt < pt_cnt) break;
            pt_cnt += idx_batch_cnt[k];
            bs_idx = k;
        }
        int start = 0;
        for (int k = 0; k < bs_idx; k++) start += features_batch_cnt[k];
        const float *f = features + (size_t)start * C;
        for (int s = 0; s < nsample; ++s) {
            int id = idx[(size_t)pt * nsample + s];
            if (id < 0) continue; /* :88 */
            for (int c = 0; c < C; ++c)
                out[(size_t)pt * C * nsample + (size_t)c * nsample + s] = f[(size_t)id * C + c];
        }
    }
    return 0;
}

/* K6  group_features_grad_kernel_stack: group_features_gpu.cu:15-47.
 * Sequential sum order (pt, c, s ascending) replaces the atomicAdd order.   */
int orc_group_features_grad(int B, int M, int C, int N, int nsample,
                            const float *grad_out, const int *idx,
                            const int *idx_batch_cnt,
                            const int *features_batch_cnt, float *grad_features) {
    (void)N;
    for (int pt = 0; pt < M; ++pt) {
        int bs_idx = 0, pt_cnt = idx_batch_cnt[0];
        for (int k = 1; k < B; k++) {
            if (pt < pt_cnt) break;
            pt_cnt += idx_batch_cnt[k];
            bs_idx = k;
        }
        int start = 0;
        for (int k = 0; k < bs_idx; k++) start += features_batch_cnt[k];
        for (int c = 0; c < C; ++c)
            for (int s = 0; s < nsample; ++s) {
                int id = idx[(size_t)pt * nsample + s];
                if (id < 0) continue;
                grad_features[(size_t)(start + id) * C + c] +=
                    grad_out[(size_t)pt * C * nsample + (size_t)c * nsample + s];
            }
    }
    return 0;
}

/* opt_n_threads(): pointnet2/pointnet2_batch/src/cuda_utils.h:10-14 */
int orc_opt_n_threads(int work_size) {
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

/* K7  farthest_point_sampling_kernel<block_size>:
 * pointnet2/pointnet2_batch/src/sampling_gpu.cu:100-216, with the shared-memory
 * tree of :93-98 and :149-208 simulated literally so that ties resolve exactly
 * as a CUDA block of `bs` threads would resolve them.
 * dataset (B,N,3) f32, temp (B,N) pre-filled 1e10, idxs (B,m).               */
int orc_farthest_point_sampling(int b, int n, int m, const float *dataset,
                                float *temp, int *idxs) {
    if (m <= 0) return 0;
    int bs = orc_opt_n_threads(n);
#pragma omp parallel num_threads(ORC_THREADS)
    {
    float *dists = (float *)malloc(sizeof(float) * (size_t)bs); /* one "shared memory" per thread */
    int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
#pragma omp for schedule(dynamic, 16)
    for (int bi = 0; bi < b; ++bi) {
        const float *d = dataset + (size_t)bi * n * 3;
        float *tmp = temp + (size_t)bi * n;
        int *out = idxs + (size_t)bi * m;
        int old = 0;
        out[0] = old;
        for (int j = 1; j < m; ++j) {
            float x1 = d[old * 3 + 0], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
            for (int tid = 0; tid < bs; ++tid) {
                int besti = 0;
                float best = -1;
                for (int k = tid; k < n; k += bs) {
                    float x2 = d[k * 3 + 0], y2 = d[k * 3 + 1], z2 = d[k * 3 + 2];
                    /* written as the fma chain nvcc's default -fmad=true emits
                     * (see orc_three_nn); on the path the inputs are small
                     * integers so every evaluation order is exact anyway.     */
                    float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
                    float dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    float d2 = dd < tmp[k] ? dd : tmp[k]; /* min(d, temp[k]) */
                    tmp[k] = d2;
                    besti = d2 > best ? k : besti;
                    best = d2 > best ? d2 : best;
                }
                dists[tid] = best;
                dists_i[tid] = besti;
            }
            for (int s = bs / 2; s >= 1; s /= 2) { /* :149-208 */
                for (int tid = 0; tid < s; ++tid) {
                    float v1 = dists[tid], v2 = dists[tid + s];
                    int i1 = dists_i[tid], i2 = dists_i[tid + s];
                    dists[tid] = v1 > v2 ? v1 : v2;
                    dists_i[tid] = v2 > v1 ? i2 : i1; /* :97 */
                }
            }
            old = dists_i[0];
            out[j] = old;
        }
    }
    free(dists);
    free(dists_i);
    }
    return 0;
}

/* K8  gather_points_kernel_fast: sampling_gpu.cu:15-31. points (B,C,N),
 * idx (B,M) -> out (B,C,M)                                                   */
int orc_gather_points(int b, int c, int n, int m, const float *points,
                      const int *idx, float *out) {
#pragma omp parallel for schedule(static) num_threads(ORC_THREADS)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int p = 0; p < m; ++p)
                out[((size_t)bi * c + ci) * m + p] =
                    points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + p]];
    return 0;
}

/* K11a gather_points_grad_kernel_fast: sampling_gpu.cu:53-90 */
int orc_gather_points_grad(int b, int c, int n, int m, const float *grad_out,
                           const int *idx, float *grad_points) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int p = 0; p < m; ++p)
                grad_points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + p]] +=
                    grad_out[((size_t)bi * c + ci) * m + p];
    return 0;
}

/* K9  three_nn_kernel_fast: pointnet2/pointnet2_batch/src/interpolate_gpu.cu:16-59.
 * unknown (B,N,3), known (B,M,3) -> dist2 (B,N,3), idx (B,N,3).
 * The squared distance is written as the fma chain nvcc's default
 * -fmad=true produces for `a*a + b*b + c*c` (setup.py passes no nvcc flags);
 * explicit fmaf() keeps the oracle and the HIP kernel bit-identical.        */
int orc_three_nn(int b, int n, int m, const float *unknown, const float *known,
                 float *dist2, int *idx) {
#pragma omp parallel for schedule(static) num_threads(ORC_THREADS)
    for (int bi = 0; bi < b; ++bi)
        for (int p = 0; p < n; ++p) {
            const float *u = unknown + ((size_t)bi * n + p) * 3;
            const float *kn = known + (size_t)bi * m * 3;
            float ux = u[0], uy = u[1], uz = u[2];
            double best1 = 1e40, best2 = 1e40, best3 = 1e40;
            int besti1 = 0, besti2 = 0, besti3 = 0;
            for (int k = 0; k < m; ++k) {
                float dx = ux - kn[k * 3 + 0];
                float dy = uy - kn[k * 3 + 1];
                float dz = uz - kn[k * 3 + 2];
                float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                if (d < best1) {
                    best3 = best2; besti3 = besti2;
                    best2 = best1; besti2 = besti1;
                    best1 = d; besti1 = k;
                } else if (d < best2) {
                    best3 = best2; besti3 = besti2;
                    best2 = d; besti2 = k;
                } else if (d < best3) {
                    best3 = d; besti3 = k;
                }
            }
            float *o = dist2 + ((size_t)bi * n + p) * 3;
            int *oi = idx + ((size_t)bi * n + p) * 3;
            o[0] = (float)best1; o[1] = (float)best2; o[2] = (float)best3;
            oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
        }
    return 0;
}

/* K10 group_points_kernel_fast: pointnet2/pointnet2_batch/src/group_points_gpu.cu:53-72.
 * points (B,C,N), idx (B,npoints,nsample) -> out (B,C,npoints,nsample)       */
int orc_group_points(int b, int c, int n, int npoints, int nsample,
                     const float *points, const int *idx, float *out) {
#pragma omp parallel for schedule(static) num_threads(ORC_THREADS)
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int p = 0; p < npoints; ++p)
                for (int s = 0; s < nsample; ++s)
                    out[(((size_t)bi * c + ci) * npoints + p) * nsample + s] =
                        points[((size_t)bi * c + ci) * n +
                               idx[((size_t)bi * npoints + p) * nsample + s]];
    return 0;
}

/* K11b group_points_grad_kernel_fast: group_points_gpu.cu:14-50 */
int orc_group_points_grad(int b, int c, int n, int npoints, int nsample,
                          const float *grad_out, const int *idx,
                          float *grad_points) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int p = 0; p < npoints; ++p)
                for (int s = 0; s < nsample; ++s)
                    grad_points[((size_t)bi * c + ci) * n +
                                idx[((size_t)bi * npoints + p) * nsample + s]] +=
                        grad_out[(((size_t)bi * c + ci) * npoints + p) * nsample + s];
    return 0;
}
