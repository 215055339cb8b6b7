/*
 * mssvt_hip.h -- C ABI of libmssvt_hip.so: the MI355X (gfx950) implementation of
 * MsSVT's mixed-scale sparse-voxel attention hot path.
 *
 * This is the drop-in boundary.  Every entry point in
 * part 1 replaces one pybind wrapper of the reference's two CUDA extensions,
 * with the same argument order and meaning, except that
 *   - at::Tensor arguments become raw DEVICE pointers (caller-owned buffers,
 *     C-contiguous, pre-allocated and pre-filled exactly as the reference's
 *     Python callers pre-fill them);
 *   - a trailing `void *stream` (a hipStream_t; NULL = default stream) is added:
 *     the reference launches on the legacy default stream;
 *   - a few calls take a caller-provided `int *workspace` (sizes below) -- the
 *     library never allocates, keeps no global state and never exit()s;
 *   - the return value is 0 on success, a positive hipError_t if a launch
 *     failed, or a negative MSSVT_E_* code for argument errors (the reference
 *     returns the constant 1 and exit(-1)s on error).
 * All pointers are device pointers unless stated otherwise.  All kernels are
 * asynchronous with respect to the host.
 *
 * "ref:" citations are relative to /root/reference/pcdet/ops/.
 */
#ifndef MSSVT_HIP_H
#define MSSVT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MSSVT_OK 0
#define MSSVT_E_BADARG (-1)   /* null pointer / non-positive size            */
#define MSSVT_E_TOOLARGE (-2) /* a size exceeds what the kernel supports     */

/* Library / build identification.  Returns e.g. 100 for 1.0.0. */
int mssvt_hip_abi_version(void);
/* Static string for a status code returned by any function below. */
const char *mssvt_hip_status_string(int status);

/* ======================================================================== *
 * Part 1a -- replacements for pybind module `mssvt_ops_cuda`
 *            (ref: mssvt/src/ms_api.cpp:7-14)
 * ======================================================================== */

/* Number of int32 words of workspace the two hash builders need for
 * `num_voxels` voxels (flags, ranks and scan partials). */
long long mssvt_hash_workspace_ints(int num_voxels, int batch_size);

/* ref: build_mapping_with_hash_wrapper, mssvt/src/ms_sparse_attention.cpp:23-35
 *      (kernel ms_sparse_attention_gpu.cu:66-97).
 * v_indices (N,4) int32 [b,z,y,x], batch-contiguous; v_bs_cnt (B) int32;
 * xyz_to_vidx (B,H,2) int32 pre-filled with -1 by the caller.
 * The table layout produced equals SEQUENTIAL insertion in voxel-index order
 * (one legal outcome of the reference's racing atomicCAS insertions); for
 * duplicate keys the highest voxel index owns the value.
 * Extra vs the reference: batch_size (to bound b), workspace, stream.        */
int mssvt_build_mapping_with_hash(int x_max, int y_max, int z_max, int num_voxels, int hash_size,
                                  int batch_size, const int *v_indices, const int *v_bs_cnt,
                                  int *xyz_to_vidx, int *workspace, void *stream);

/* ref: window_with_hash_wrapper, ms_sparse_attention.cpp:37-59 (kernel :117-168).
 * w_indices (B,num_windows,3) int32 pre-filled -1 -> rows [wz,wy,wx];
 * xyz_to_vidx (B,H,2) pre-filled -1; vcount (B) zeroed.
 * Windows are numbered by FIRST OCCURRENCE in voxel-index order within each
 * sample and the table equals sequential insertion in that order.  Windows
 * beyond num_windows are counted in vcount but not written (the reference
 * writes out of bounds there).                                               */
int mssvt_window_with_hash(int x_wgs, int y_wgs, int z_wgs, int x_ws, int y_ws, int z_ws,
                           int num_voxels, int num_windows, int hash_size, int batch_size,
                           const int *v_indices, int *w_indices, int *xyz_to_vidx, int *vcount,
                           int *workspace, void *stream);

/* ref: gather_two_window_voxels_with_hash_wrapper, ms_sparse_attention.cpp:61-120
 *      (kernel :193-350).  vox_ind_* (nw,max_num_*) pre-filled -1, vox_coord_*
 * (nw,max_num_*,3) pre-filled 0; vox_query_* (num_*,3) int32 offset tables;
 * win_indices (nw,4) [b,wz,wy,wx]; xyz_to_vidx (B,H,2).                      */
int mssvt_gather_two_window_voxels_with_hash(
    int x_max, int y_max, int z_max, int x_ws, int y_ws, int z_ws, int max_num_odd,
    int max_num_even, int max_num_win1, int max_num_win2, int num_wins, int hash_size, int num_odd,
    int num_even, int num_win1, int num_win2, int *vox_ind_odd, int *vox_ind_even,
    int *vox_ind_win1, int *vox_ind_win2, int *vox_coord_odd, int *vox_coord_even,
    int *vox_coord_win1, int *vox_coord_win2, const int *vox_query_odd, const int *vox_query_even,
    const int *vox_query_win1, const int *vox_query_win2, const int *win_indices,
    const int *xyz_to_vidx, void *stream);

/* ref: gather_one_window_voxels_with_hash_wrapper, ms_sparse_attention.cpp:122-150
 *      (kernel :383-433).                                                    */
int mssvt_gather_one_window_voxels_with_hash(int x_max, int y_max, int z_max, int x_ws, int y_ws,
                                             int z_ws, int max_num_win1, int num_wins,
                                             int hash_size, int num_win1, int *vox_ind_win1,
                                             int *vox_coord_win1, const int *vox_query_win1,
                                             const int *win_indices, const int *xyz_to_vidx,
                                             void *stream);

/* ref: group_features_wrapper_stack, mssvt/src/group_features.cpp:50-68
 *      (kernel group_features_gpu.cu:73-106).  features (N,C) f32, idx (M,nsample)
 * int32 (index within the sample, <0 = skip), out (M,C,nsample) f32 pre-zeroed. */
int mssvt_group_features(int B, int M, int C, int nsample, const float *features,
                         const int *features_batch_cnt, const int *idx, const int *idx_batch_cnt,
                         float *out, void *stream);

/* ref: group_features_grad_wrapper_stack, group_features.cpp:29-47
 *      (kernel group_features_gpu.cu:15-47).  grad_features (N,C) pre-zeroed.  */
int mssvt_group_features_grad(int B, int M, int C, int N, int nsample, const float *grad_out,
                              const int *idx, const int *idx_batch_cnt,
                              const int *features_batch_cnt, float *grad_features, void *stream);

/* ======================================================================== *
 * Part 1b -- replacements for the four forward ops (and two grads) of pybind
 *            module `pointnet2_batch_cuda` that the path uses
 *            (ref: pointnet2/pointnet2_batch/src/pointnet2_api.cpp:10-24)
 * ======================================================================== */

/* ref: farthest_point_sampling_wrapper, pointnet2/pointnet2_batch/src/sampling.cpp:41-50
 *      (kernel sampling_gpu.cu:100-216).  dataset (B,N,3) f32, temp (B,N) f32
 * pre-filled 1e10 (scratch, overwritten), idxs (B,m) int32.  Ties resolve
 * exactly as the reference block of opt_n_threads(N) threads resolves them.   */
int mssvt_farthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                  int *idxs, void *stream);

/* ref: gather_points_wrapper_fast, sampling.cpp:18-27 (kernel sampling_gpu.cu:15-31).
 * points (B,C,N), idx (B,npoints) -> out (B,C,npoints).                       */
int mssvt_gather_points(int b, int c, int n, int npoints, const float *points, const int *idx,
                        float *out, void *stream);
/* ref: gather_points_grad_wrapper_fast, sampling.cpp:29-39 (kernel :53-90).   */
int mssvt_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                             const int *idx, float *grad_points, void *stream);

/* ref: three_nn_wrapper_fast, pointnet2/pointnet2_batch/src/interpolate.cpp:21-30
 *      (kernel interpolate_gpu.cu:16-59).  unknown (B,N,3), known (B,M,3) ->
 * dist2 (B,N,3) squared distances, idx (B,N,3).                               */
int mssvt_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                   int *idx, void *stream);

/* ref: group_points_wrapper_fast, pointnet2/pointnet2_batch/src/group_points.cpp:30-40
 *      (kernel group_points_gpu.cu:53-72).  points (B,C,N), idx (B,npoints,nsample)
 * -> out (B,C,npoints,nsample).                                               */
int mssvt_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                       const int *idx, float *out, void *stream);
/* ref: group_points_grad_wrapper_fast, group_points.cpp:18-28 (kernel :14-50). */
int mssvt_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                            const int *idx, float *grad_points, void *stream);

/* ======================================================================== *
 * Part 2 -- device-resident variants used by the module fast path.  They
 *           compute exactly what the reference's Python glue computes around
 *           the wrappers above, without host round trips.
 * ======================================================================== */

/* workspace[0] = status bits after any hash builder call (host-readable after a
 * stream sync): */
#define MSSVT_ST_DUPLICATE_KEY 1   /* duplicate voxel coordinates were present   */
#define MSSVT_ST_TABLE_OVERFLOW 2  /* a sample has more distinct keys than H     */
#define MSSVT_ST_WINDOW_OVERFLOW 4 /* a sample has more than max_num_wins windows */

/* K2 + the per-sample compaction of ref mssvt/mssvt_ops.py:45-53 on the device:
 * win_ind receives the (nw,4) rows [b,wz,wy,wx] in sample order, then window
 * number order (capacity: num_voxels rows); vcount (B) = windows per sample
 * (= k_bs_cnt of ref mssvt_backbone.py:218); workspace[1] = nw.              */
int mssvt_window_partition_compact(int x_wgs, int y_wgs, int z_wgs, int x_ws, int y_ws, int z_ws,
                                   int num_voxels, int max_num_wins, int hash_size, int batch_size,
                                   const int *v_indices, int *win_ind, int *xyz_to_vidx,
                                   int *vcount, int *workspace, void *stream);

/* Several window partitions of ONE voxel list in the same five launches (grid.y = partition), e.g. a
 * Block's [3,3,5] windows and the following CompressBlock's [1,1,32] pillars: both only depend on the voxel
 * indices (ref: Block.window_partition, mssvt_backbone.py:139-152, recomputed per block there).
 * Host arrays of num_sets (<= 4) entries: win_grid3 / win_size3 (3 ints each), max_num_wins, and the
 * device pointers win_ind (N,4), tables (B,H,2; pre-filled with -1), vcount (B).  scratch_tables (array or
 * entries may be NULL): a second -1-filled (B,H,2) buffer per partition for the first-occurrence pass --
 * without it the table is its own scratch and is refilled in between.  workspaces: num_sets slices of
 * workspace_stride_ints (>= mssvt_hash_workspace_ints) ints of one allocation; slice k holds
 * [status, window count, ...] of partition k as in mssvt_window_partition_compact. */
int mssvt_window_partition_multi(int num_sets, const int *host_win_grid3, const int *host_win_size3,
                                 const int *host_max_num_wins, int num_voxels, int hash_size, int batch_size,
                                 const int *v_indices, int *const *host_win_ind, int *const *host_tables,
                                 int *const *host_scratch_tables, int *const *host_vcount, int *workspaces,
                                 long long workspace_stride_ints, void *stream);

/* Everything a resolution level needs before its first Block, in one call behind ONE fill:
 * samples' voxel counts (with_bs_cnt, mssvt_backbone.py:124-130), the voxel hash table
 * (SparseTensor.build_map_table, mssvt_utils.py:31-48 = mssvt_build_mapping_with_hash), the occupancy
 * columns of mssvt_occupancy_columns (optional: NULL) and num_sets (0..4) window partitions as in
 * mssvt_window_partition_multi.  zero_region / zero_bytes: one caller allocation that this function clears and
 * that must contain v_bs_cnt (B ints), the first 4 ints of map_workspace, occ_columns (B*X*Y words) and
 * the first 4 ints of every partition workspace (and the vcount arrays when num_voxels == 0);
 * MSSVT_E_BADARG otherwise.  map_table and the partition tables / scratch tables are pre-filled with -1 by the
 * caller as for the single entry points. */
int mssvt_level_setup(int num_voxels, int batch_size, int x_max, int y_max, int z_max, int hash_size,
                      const int *v_indices, void *zero_region, long long zero_bytes, int *v_bs_cnt, int *map_table,
                      int *map_workspace, unsigned long long *occ_columns, int num_sets, const int *host_win_grid3,
                      const int *host_win_size3, const int *host_max_num_wins, int *const *host_win_ind,
                      int *const *host_tables, int *const *host_scratch_tables, int *const *host_vcount,
                      int *workspaces, long long workspace_stride_ints, void *stream);

/* The same level set-up for a voxel list that is SORTED by (b,x,y,z) -- the order DynamicVFE emits
 * (vfe/dynamic_vfe.py:83-93,114-118: torch.unique over the voxel keys) -- from one occupancy bitmap, without the
 * voxel hash table (K1, ms_sparse_attention_gpu.cu:66-97), without the insert-min / rank passes of K2 (:117-168)
 * and without counting atomics (mssvt_utils.py:35-37), three launches behind one fill (z_max <= 64):
 *   v_bs_cnt (B), sample_start (B+1) rows of every sample; occ_columns (B*X*Y words, bit z);
 *   column_vbase (B*X*Y): voxels of the sample in earlier columns (index of cell (x,y,z) in its sample =
 *   column_vbase + popcount(word below z)); per partition k < num_sets (<= 4): win_ind[k] (num_voxels,4) window rows
 *   [b,wz,wy,wx] in first-occurrence order, tables[k] (B,H,2) pre-filled with -1 or NULL (no table wanted),
 *   vcount[k] (B) windows per sample, ws[k]: 4 header ints ([0] status bits, [1] number of windows).
 * level_status[0] gets ST_UNSORTED (8) when the list is not strictly ascending in (b,x,y,z) or holds an out-of-grid
 * voxel; every partition then reports 0 windows and the caller must use mssvt_level_setup (any order).
 * zero_region / zero_bytes: one caller allocation cleared by this call that contains sample_start, occ_columns,
 * level_status and the ws headers (zero_bytes < 0: |zero_bytes| bytes the caller has cleared itself, e.g. with
 * mssvt_fill_two -- no clear here).  scratch: mssvt_level_sorted_scratch_ints(B, X, Y) ints.                     */
long long mssvt_level_sorted_scratch_ints(int batch_size, int x_max, int y_max);
/* a[0..n_a) = value_a and b[0..n_b) = value_b (int32, both 16-byte aligned) in ONE launch: the -1 prefill of a frame's
 * tables / owner arrays (ref: torch.full(-1) per table, mssvt_utils.py:39, mssvt_ops.py:47) and its zeroed words.    */
int mssvt_fill_two(int *a, long long n_a, int value_a, int *b, long long n_b, int value_b, void *stream);
int mssvt_level_setup_sorted(int num_voxels, int batch_size, int x_max, int y_max, int z_max, int hash_size,
                             const int *v_indices, void *zero_region, long long zero_bytes, int *v_bs_cnt,
                             int *sample_start, unsigned long long *occ_columns, int *column_vbase, int *level_status,
                             int num_sets, const int *host_win_grid3, const int *host_win_size3,
                             const int *host_max_num_wins, int *const *host_win_ind, int *const *host_tables,
                             int *const *host_vcount, int *const *host_ws, int *scratch, void *stream);

/* mssvt_level_setup_sorted that also writes the K4 lists of partition `pillar_set` -- pillar windows [1,1,z], offsets of
 * vox_query_win1 (num_win1 <= 64 rows) with x = y = 0 -- i.e. the outputs of mssvt_window_plan_one(disjoint_lists = 2,
 * with_pad = 0) for that partition (k_ind (cap,max_num_win1), win_vstart, win_cnt, pair_base (cap), pair_win / pair_vox (N);
 * pair_win AND k_ind pre-filled with -1 -- only listed slots are written --; k_ind / pair_base / pair_vox may be NULL:
 * mssvt_compress_fused reads neither; table z offsets in [-32, 31]): every such window is a slab of ONE column's occupancy word, so its list falls out where
 * the window is numbered, without a launch of its own (ref gather_one_window_voxels, ms_sparse_attention_gpu.cu:383-433).
 * MSSVT_E_TOOLARGE: the partition's windows are not pillars.                                                        */
int mssvt_level_setup_sorted_pillars(
    int num_voxels, int batch_size, int x_max, int y_max, int z_max, int hash_size, const int *v_indices, void *zero_region,
    long long zero_bytes, int *v_bs_cnt, int *sample_start, unsigned long long *occ_columns, int *column_vbase,
    int *level_status, int num_sets, const int *host_win_grid3, const int *host_win_size3, const int *host_max_num_wins,
    int *const *host_win_ind, int *const *host_tables, int *const *host_vcount, int *const *host_ws, int *scratch,
    int pillar_set, int max_num_win1, int num_win1, const int *vox_query_win1, int *k_ind, int *win_vstart, int *win_cnt,
    int *pair_base, int *pair_win, int *pair_vox, void *stream);

/* Fused window plan of a two-scale Block: K3 + 2 x K7 + 2 x K8 + the key-mask logic
 * of ref mssvt_backbone.py:247-258 in one launch, one wavefront per window, hit lists
 * kept in LDS.  num_wins_dev: DEVICE scalar (e.g. workspace+1 of
 * mssvt_window_partition_compact); win_capacity: rows allocated in the per-window
 * outputs.  Outputs: ind_odd/ind_even/ind_win1 (cap,max_num_*) as K3 writes them (-1
 * padded); k_ind1/k_ind2 (cap,K) sampled key voxels of the win1 / win2 list (an EMPTY
 * slot picked by FPS becomes voxel 0, as in the reference); k_mask1/k_mask2 (cap,K)
 * bytes, 1 = masked; win_vstart (cap) first feature row of the window's sample;
 * owner_* (N) pre-filled -1: highest flat list slot (w*max_num + s) holding the voxel.
 * Optional resolved metadata (kmeta1 == NULL: skipped; each qmeta_* may be NULL on its own): qmeta_* (cap,max_num_*,4), kmeta1/2
 * (cap,K,4) f32 = (voxel centre - window centre in metres, bits of the global feature row or
 * -1 for empty / masked slots); wcentre (cap,4) = window centre; nq_valid (3,cap) = valid odd /
 * even / win1 entries per window.  indices (N,4) voxel coords.  With the metadata asked for, the list rows ind_*, the key
 * indices / masks k_ind* / k_mask* and the owner_* arrays may each be NULL (the fused consumers read the metadata and the
 * interpolation tables: mssvt_frame_forward passes none of them); the weights of the tables use the hardware's square root
 * and reciprocal (1 ulp each).
 * column_vbase / level_status_dev (optional, with occ_columns; from mssvt_level_setup_sorted): for a voxel list sorted
 * by (b,x,y,z) the index of an occupied cell is column_vbase + popcount(column word below z) and the hash is not
 * probed at all (xyz_to_vidx may then be NULL); when level_status_dev[0] has ST_UNSORTED (8) set the hash is used.
 * win_counts_dev (optional, B ints: windows per sample, as the window partition reports them): the windows are
 * accepted and unused (a centre-out work order was measured: no gain).
 * num_tabs (0..4) interpolation tables built in the same launch (what mssvt_block_interp_table_multi builds from the
 * finished lists; requires kmeta1 and window lists that cannot overlap, i.e. every voxel owned by one window -- the
 * caller checks the tables): host_tab_list[t] = query list (0 odd, 1 even, 2 win1), host_tab_interp[t] = the Block's
 * use_feature_interpolation, host_tab_zero_row[t] = a row of the attention buffer that holds zeros, host_tab_row[t]
 * (N,4) int32 pre-filled with -1 / host_tab_w[t] (N,4) f32 as for mssvt_block_interp_table.                    */
int mssvt_window_plan_two(
    int x_max, int y_max, int z_max, int x_ws, int y_ws, int z_ws, int max_num_odd, int max_num_even,
    int max_num_win1, int max_num_win2, int hash_size, int batch_size, int num_odd, int num_even,
    int num_win1, int num_win2, const int *vox_query_odd, const int *vox_query_even,
    const int *vox_query_win1, const int *vox_query_win2, int key_num_sample, const int *win_indices,
    const int *num_wins_dev, int win_capacity, const int *xyz_to_vidx, const int *v_bs_cnt,
    int *ind_odd, int *ind_even, int *ind_win1, int *k_ind1, int *k_ind2, unsigned char *k_mask1,
    unsigned char *k_mask2, int *win_vstart, int *owner_win1, int *owner_odd, int *owner_even,
    const int *indices, const float *host_voxel_size3, const float *host_range_min3,
    const float *host_win_size3, float *qmeta_odd, float *qmeta_even, float *qmeta_win1, float *kmeta1,
    float *kmeta2, float *wcentre, int *nq_valid, const unsigned long long *occ_columns,
    const int *host_footprint4, const int *packed_offsets, const int *column_vbase, const int *level_status_dev,
    const int *win_counts_dev, int num_tabs, const int *host_tab_list, const int *host_tab_interp,
    const int *host_tab_zero_row, int *const *host_tab_row, float *const *host_tab_w, void *stream);

/* Occupancy columns of a voxel set (z_max <= 64): columns (B*x_max*y_max) 64-bit words, bit z of
 * word (b*x_max + x)*y_max + y set when cell (b,x,y,z) holds a voxel.  Optional input of
 * mssvt_window_plan_two (together with host_footprint4 = {min x offset, min y offset, x extent,
 * y extent} of the four query tables and packed_offsets = the tables concatenated odd | even |
 * win1 | win2, one word (x+64) | (y+64)<<7 | (z+64)<<14 | column<<21 per offset, column = (x - min x) * y extent +
 * (y - min y), on the device; |offset| <= 60, x extent * y extent <= 1024): the K3 hit test then reads one word per (x,y) column of the
 * neighbourhood instead of probing the hash for each of its cells (ref K3 probes all of them,
 * ms_sparse_attention_gpu.cu:193-330); the hash is only probed for the hits.               */
int mssvt_occupancy_columns(const int *indices, int num_voxels, int batch_size, int x_max, int y_max,
                            int z_max, unsigned long long *columns, void *stream);

/* Work order and compact query rows for mssvt_block_attention_group, from one row of the
 * plan's (3,cap) nq_valid (odd / even / win1) and that list's qmeta (cap,nq,4):
 *   perm (cap)          the windows with >= 1 valid query, sorted by descending nq_valid;
 *   num_active_dev      how many;
 *   q_off (cap)         exclusive prefix sum of nq_valid in window order = first compact
 *                       query row of each window;
 *   qrow_meta (rows,4)  per compact row: the slot's qmeta entry (rel.xyz, bits(feature row));
 *   qrow_src (rows,2)   per compact row: (window, attn row = window * nq + slot);
 *   num_rows_dev        total rows (<= num_voxels: the query lists of one pattern are
 *                       disjoint); rows beyond row_capacity are dropped.                  */
int mssvt_plan_order(const int *num_wins_dev, const int *nq_valid, int nq, const float *qmeta,
                     int win_capacity, int row_capacity, int *perm, int *num_active_dev, int *q_off,
                     float *qrow_meta, int *qrow_src, int *num_rows_dev, void *stream);

/* mssvt_plan_order for several query lists of one plan in ONE launch pair (host arrays of
 * num_sets <= 4 device pointers / sizes, one entry per list).                                 */
int mssvt_plan_order_multi(int num_sets, const int *num_wins_dev, const int *const *host_nq_valid,
                           const int *host_nq, const float *const *host_qmeta, int win_capacity,
                           int row_capacity, int *const *host_perm, int *const *host_num_active,
                           int *const *host_q_off, float *const *host_qrow_meta, int *const *host_qrow_src,
                           int *const *host_num_rows, void *stream);

/* Fused attention of a Block, all head groups (group g = channels [c0[g], c0[g]+Cg[g]),
 * Cg = heads[g]*head_dim <= 64, attends to key scale g): gathers + positional MLP +
 * MixedScaleAttention (ref mssvt_backbone.py:260-295, mssvt_utils.py:112-150) for every valid
 * query of every window.  xhat (N,C) = norm1 output; perm / num_active_dev / q_off / qrow_meta /
 * qrow_src / num_rows_dev: from mssvt_plan_order for the query list of this block's cbs_pattern,
 * nq_valid = the row it was built from; host_kmeta[g] (cap,K,4) / wcentre (cap,4): the plan
 * kernel's resolved metadata of key scale g; host_Wq[g] (Cg,Cg), host_Wkv[g] (2Cg,Cg),
 * host_Wo[g] (Cg,Cg) + biases, Wpos (C,6), bpos = the module's parameters (host arrays of device
 * pointers); qbuf: scratch of row_capacity x sum_g 4*ceil(heads[g]/4)*Cg[g] floats handed between
 * the three launches (queries / keys+softmax / output); attn (cap*nq [+1],C): rows of valid
 * query slots are written.  Groups of equal width share the launches (grid.y = group).
 * C and every c0 must be multiples of 4.                                                  */
int mssvt_block_attention(
    int C, int num_groups, const int *host_c0, const int *host_cg, const int *host_heads, int head_dim, float scale,
    int nq, int key_num_sample, const float *xhat, const int *num_active_dev, const int *perm, const int *q_off,
    const int *nq_valid, const int *num_rows_dev, int row_capacity, const float *qrow_meta, const int *qrow_src,
    const float *const *host_kmeta, const float *wcentre, const float *const *host_Wq, const float *const *host_bq,
    const float *const *host_Wkv, const float *const *host_bkv, const float *const *host_Wo,
    const float *const *host_bo, const float *Wpos, const float *bpos, float *qbuf, float *attn, void *stream);

/* mssvt_block_attention with split-fp16 matrix operands: every product sum as three v_mfma_f32_16x16x32_f16 on
 * (hi, lo = 2^11 (v - hi)) fp16 halves with fp32 accumulation -- the fp32 instruction's error at ~1/4 of its cycles.
 * The per-window launch (k_attn_kvh: scores, weighted key sum; Qt crosses qbuf pre-split, same bytes) always; the
 * two row-tiled launches (k_attn_q16 / k_attn_o16) when host_packed gives one mssvt_attn_pack_weights blob per head
 * group (head_dim 16), else they keep the fp32 instruction (host_packed may be NULL).  With the blobs and
 * key_num_sample <= 32 only Q' crosses qbuf (a quarter of the bytes) and the window launch forms Qt = (scale Wk_h)^T q'_h
 * itself from the blob's head-pair fragments, its softmax in base 2 (log2 e folded into them).  Same arguments and results (to
 * the fp32 tolerance) as mssvt_block_attention; shapes outside Cg % 32 == 0, 16 < key_num_sample <= 64 run the fp32
 * form.  The CALLER guarantees the fp16 range of key tokens, Q', Qt, Xbar, V (mssvt_amd/fused.py, _attn_kv16_ok). */
int mssvt_block_attention_kv16(
    int C, int num_groups, const int *host_c0, const int *host_cg, const int *host_heads, int head_dim, float scale,
    int nq, int key_num_sample, const float *xhat, const int *num_active_dev, const int *perm, const int *q_off,
    const int *nq_valid, const int *num_rows_dev, int row_capacity, const float *qrow_meta, const int *qrow_src,
    const float *const *host_kmeta, const float *wcentre, const float *const *host_Wq, const float *const *host_bq,
    const float *const *host_Wkv, const float *const *host_bkv, const float *const *host_Wo,
    const float *const *host_bo, const float *Wpos, const float *bpos, float *qbuf, float *attn,
    const void *const *host_packed, void *stream);

/* Split-fp16 fragments of one head group's projections for mssvt_block_attention_kv16, in MFMA operand order: Wq
 * (Cg,Cg), Wkv (2Cg,Cg: K rows then V rows; the softmax scale folded into the K fragments, scale x log2 e into their
 * head-pair copy), Wo (Cg,Cg) -- nn.Linear layouts of ref mssvt_utils.py:92-103.  Once per parameter version.  mssvt_attn_packed_bytes: size of `packed`, 0 = shape not
 * instantiated (head_dim 16, Cg 32 / 64).                                                                          */
long long mssvt_attn_packed_bytes(int Cg, int head_dim);
int mssvt_attn_pack_weights(int Cg, int head_dim, float scale, const float *Wq, const float *Wkv, const float *Wo,
                            void *packed, void *stream);

/* mssvt_block_attention with bf16 matrix-core operands (BASELINE configs[2]; an extension of this build -- the
 * reference's MixedScaleAttention computes in fp32, ref mssvt_utils.py:112-150): ONE launch, one wavefront per
 * (window, head group); keys and values are projected inside the kernel (v_mfma_f32_16x16x32_bf16), so there is
 * no qbuf hand-off.  Tokens, weights, Q', K', V', P and O are rounded to bf16 as MFMA operands; accumulation,
 * softmax, biases, the positional MLP and the attention rows written stay fp32.  Same arguments as
 * mssvt_block_attention minus qbuf.  MSSVT_E_TOOLARGE: shape not instantiated (use the fp32 entry point).   */
int mssvt_block_attention_bf16(
    int C, int num_groups, const int *host_c0, const int *host_cg, const int *host_heads, int head_dim, float scale,
    int nq, int key_num_sample, const float *xhat, const int *num_active_dev, const int *perm, const int *q_off,
    const int *nq_valid, const int *num_rows_dev, int row_capacity, const float *qrow_meta, const int *qrow_src,
    const float *const *host_kmeta, const float *wcentre, const float *const *host_Wq, const float *const *host_bq,
    const float *const *host_Wkv, const float *const *host_bkv, const float *const *host_Wo,
    const float *const *host_bo, const float *Wpos, const float *bpos, float *attn, void *stream);

/* 3-NN inverse-distance interpolation of the attention rows onto the win1 voxels (K9,
 * K10, ref mssvt_backbone.py:298-311) + scatter + first residual (ref :313-338):
 * x_new[v] = interp(v) + x_in[v] for every voxel v owned by a list slot; rows of other
 * voxels are left untouched (the caller pre-fills x_new = 2*x_in, ref quirk R12).
 * use_interpolation = 0: x_new[v] = attn[slot] + x_in[v] for the query voxels only.   */
int mssvt_block_interp_scatter(int C, int nq, int n_upd, int use_interpolation, const float *attn,
                               const float *x_in, float *x_new, const int *indices,
                               const int *win_ind, const int *num_wins_dev, int win_capacity,
                               const int *win_vstart, const int *q_ind, const int *upd_ind,
                               const int *owner, const float *host_voxel_size3,
                               const float *host_range_min3, void *stream);

/* ---- CompressBlock fast path (ref mssvt_backbone.py:351-398): ragged pieces ---------
 * K4 list per window + allocation of one "pair row" per valid (window, slot) (+ one PAD
 * row per window if with_pad).  k_ind (cap,max_num_win1) as K4 writes it; win_cnt (cap)
 * valid slots; pair_base (cap); pair_win / pair_vox (row capacity) window id and global
 * voxel row (-1 = pad) of every pair row.  disjoint_lists != 0 (no voxel in two lists: odd
 * window sizes): pair row = the voxel's feature row, pad row of window w = num_voxels + w,
 * pair_win must be pre-filled with -1, pair_base = -1.  Otherwise rows are reserved through
 * counters[0] (= rows handed out; cleared by this call) and pair_base[w] is the window's
 * first row.  With disjoint lists counters[0] is not touched.
 * occ_columns / column_vbase / level_status_dev (optional, all or none; mssvt_level_setup_sorted): sorted voxel
 * list -> occupancy bit = hit, column base + popcount below z = voxel index, the hash is not probed
 * (xyz_to_vidx may be NULL) unless level_status_dev[0] has ST_UNSORTED (8) set.
 * disjoint_lists == 2: disjoint AND every table offset has x = y = 0 (the caller checked): with x_ws = y_ws = 1 on a
 * sorted level the lists are built one LANE per window from the column words (pillar windows).        */
int mssvt_window_plan_one(int x_max, int y_max, int z_max, int x_ws, int y_ws, int z_ws,
                          int max_num_win1, int hash_size, int num_win1, const int *vox_query_win1,
                          const int *win_indices, const int *num_wins_dev, int win_capacity,
                          const int *xyz_to_vidx, const int *v_bs_cnt, int with_pad,
                          int disjoint_lists, int num_voxels, int *k_ind, int *win_vstart, int *win_cnt,
                          int *pair_base, int *pair_win, int *pair_vox, int *counters,
                          const unsigned long long *occ_columns, const int *column_vbase,
                          const int *level_status_dev, void *stream);
/* out (R,C) = relu(W1 [voxel centre - window centre ; window centre] + b1) per pair row
 * (first layer of pos_proj, ref :49-54, :372-373).                                     */
int mssvt_compress_pos1(int C, const int *num_rows_dev, int row_capacity, const int *pair_win,
                        const int *pair_vox, const int *indices, const int *win_indices,
                        const float *host_voxel_size3, const float *host_range_min3,
                        const float *host_win_size3, const float *W1, const float *b1, float *out,
                        void *stream);
/* rows[r] += xhat[pair_vox[r]] for non-pad rows (key token = feature + positional emb.). */
int mssvt_compress_add_features(int C, const int *num_rows_dev, int row_capacity, const int *pair_vox,
                                const float *xhat, float *rows, void *stream);
/* q_tok (nw,C) = channel-wise max over the window's zero-padded key features (ref :370). */
int mssvt_compress_pool(int C, int max_num_win1, const int *num_wins_dev, int win_capacity,
                        const int *k_ind, const int *win_vstart, const int *win_cnt, const float *xhat,
                        float *q_tok, void *stream);
/* nq=1 attention of head group `group` (channels [c0,c0+Cg), list slots
 * [group*keys_per_group, (group+1)*keys_per_group)): qp (nw,C) projected queries,
 * kv (R,2*Cg) = [K|V] rows of this group's to_kvs for every pair row -> out (nw,C)
 * columns [c0,c0+Cg) (before the output projection).                                   */
int mssvt_compress_attention_group(int C, int c0, int Cg, int head_dim, float scale,
                                   int keys_per_group, int group, int with_pad, int max_num_win1,
                                   int num_voxels, const int *num_wins_dev, int win_capacity,
                                   const int *win_cnt, const int *pair_base, const int *k_ind,
                                   const int *win_vstart, const float *qp, const float *kv, float *out,
                                   void *stream);

/* CompressBlock attention up to (not including) the FFN tail, for ONE head group and
 * non-overlapping window lists (ref mssvt_backbone.py:351-383; MixedScaleAttention
 * mssvt_utils.py:112-150), four launches on the fp32 matrix cores, every count read on the
 * device (no host synchronisation):  out (nw,C) = projs( softmax_v( scale (Wq max_v xhat_v + bq)
 * . (Wk k_v + bk) ) (Wv k_v + bv) ),  k_v = xhat_v + pos_proj([centre_v - centre_w ; centre_w]).
 * k_ind / win_vstart / win_cnt / pair_win: from mssvt_window_plan_one with disjoint_lists = 1,
 * with_pad = 0.  Wpos1 (C,6), Wpos2 (C,C), Wq (C,C), Wkv (2C,C), Wo (C,C) + biases: the module's
 * parameters.  Scratch: qp (win_capacity,C), ktok (N,C), score (N,C/head_dim), vp (N,C).
 * split_f16 != 0: the C x C products with every fp32 operand split into two fp16 halves (22 of 24 mantissa bits)
 * (3 x v_mfma_f32_16x16x32_f16, fp32 accumulation: the fp32 instruction's error at 3/16 of its
 * cycles); the CALLER guarantees the fp16 range of xhat, the positional hidden layer, the key
 * tokens and the V rows (mssvt_amd/fused.py bounds them from the parameters), else pass 0.
 * Instantiated for C in {32,64,128}, head_dim in {8,16,32} (C/head_dim <= 8).               */
int mssvt_compress_fused(
    int C, int head_dim, float scale, int max_num_win1, int num_voxels, const int *num_wins_dev, int win_capacity,
    const int *win_ind, const int *indices, const int *k_ind, const int *win_vstart, const int *win_cnt,
    const int *pair_win, const float *host_voxel_size3, const float *host_range_min3, const float *host_win_size3,
    const float *xhat, const float *Wpos1, const float *bpos1, const float *Wpos2, const float *bpos2,
    const float *Wq, const float *bq, const float *Wkv, const float *bkv, const float *Wo, const float *bo,
    float *qp, float *ktok, float *score, float *vp, float *out, int split_f16, void *stream);

/* The same attention (same inputs, same `out`) in ONE launch and without any scratch, for the case the detector runs:
 * a level set up as SORTED (mssvt_level_setup_sorted*: windows are numbered in row order), pillar windows x_ws = y_ws = 1
 * (window cell = (x, y, z / z_ws): the window rows are not read) whose lists cannot be truncated (z_ws <= max_num_win1 <= 32),
 * so that every window is one run of consecutive voxel rows and
 * consecutive windows are consecutive runs; one head group, head_dim 16, C = 128; split-fp16 products (the caller
 * guarantees the fp16 range exactly as for mssvt_compress_fused(split_f16 = 1)).  One workgroup per CU keeps pos_proj.2, Wk
 * and Wv as MFMA fragments in registers (wave h = head h), Wq / Wo in LDS; the key tokens, K, V, the scores and the
 * projected queries never leave the CU (csrc/compress_ws.hip).  packed: mssvt_compress_ws_packed_bytes(C) bytes written by
 * mssvt_compress_ws_pack from pos_proj.2 (C,C), to_q (C,C), to_kv (2C,C), proj (C,C) -- once per parameter version.
 * Of the K4 plan it reads win_cnt and pair_win only (the lists themselves are runs of rows).
 * MSSVT_E_TOOLARGE: shape not covered (use mssvt_compress_fused).  Deterministic; differs from mssvt_compress_fused by the
 * association of the softmax sums only.                                                                              */
long long mssvt_compress_ws_packed_bytes(int C);
int mssvt_compress_ws_pack(int C, const float *Wpos2, const float *Wq, const float *Wkv, const float *Wo, void *packed,
                           void *stream);
int mssvt_compress_ws(int C, int head_dim, float scale, int z_ws, int max_num_win1, int num_voxels, const int *num_wins_dev,
                      int win_capacity, const int *indices, const int *win_cnt, const int *pair_win,
                      const float *host_voxel_size3, const float *host_range_min3, const float *host_win_size3,
                      const float *xhat, const float *Wpos1, const float *bpos1, const float *bpos2, const float *bq,
                      const float *bkv, const float *bo, const void *packed, float *out, void *stream);

/* Backward of mssvt_layer_norm (training path; autograd's LayerNorm backward in the reference): dx (N,C), dweight (C),
 * dbias (C) from x, dy; mean / rstd are recomputed.  The column sums are per-workgroup partial rows in `workspace`
 * (512 * 2 * C floats) added in workgroup order: deterministic.  C in {16,32,64,128,256}.                      */
int mssvt_layer_norm_backward(const float *x, const float *dy, int num_rows, int C, const float *weight, float eps,
                              float *dx, float *dweight, float *dbias, float *workspace, void *stream);
/* The same with dx = (LayerNorm backward of dy) + dres: dres (N,C) or NULL = the gradient that reaches x through its
 * other uses (the residual connection around the normalised branch: ref mssvt_backbone.py:241, :339-343) -- the add
 * autograd would run as a pass of its own.                                                                       */
int mssvt_layer_norm_backward_residual(const float *x, const float *dy, const float *dres, int num_rows, int C,
                                       const float *weight, float eps, float *dx, float *dweight, float *dbias,
                                       float *workspace, void *stream);

/* Rows per sample of a (N,4) [b,z,y,x] int32 index tensor -> counts (B) int32, on the device
 * (ref: the host loops with .item() of mssvt_utils.py:35-37 / mssvt_backbone.py:124-130).   */
int mssvt_batch_counts(const int *indices, int num_rows, int batch_size, int *counts, void *stream);

/* y = LayerNorm(x) over the last dimension (N,C) f32, C in {16,32,64,128,256}
 * (ref: norm1, mssvt_backbone.py:241); MSSVT_E_TOOLARGE otherwise.                          */
int mssvt_layer_norm(const float *x, int num_rows, int C, const float *weight, const float *bias,
                     float eps, float *y, void *stream);

/* Fused feed-forward tail of a block on the fp32 matrix cores (ref mssvt_backbone.py:336-343,
 * :383-387): x = owner && owner[v] < 0 ? 2*x_in[v] : x_new[v];
 * y = x + linear2(relu(linear1(norm(x)))); optionally y_norm = next_norm(y) (the next block's
 * norm1).  x_new/x_in/y/y_norm (N,C) f32; W1 (FF,C), W2 (C,FF) as in nn.Linear.
 * phases 3 (fp32 matrix instruction): hidden = scratch of n_rows x FF floats, two launches with LDS-resident
 * weights (GEMM1 | GEMM2, the hidden activations make one round trip through `hidden`; hidden NULL ->
 * MSSVT_E_BADARG).  num_rows_dev (optional): the row count is read on the device and n_rows is only the
 * capacity.  phases 1 = only the first launch (LayerNorm + GEMM1 + ReLU -> hidden, x parked in y), 2 = only
 * the second (GEMM2 + residual + next norm) -- for measurement.
 * phases 4 (num_rows_dev allowed; hidden = NULL, or the fragments written by
 * mssvt_ffn_pack_weights for these W1 / W2: saves the in-kernel split): ONE launch with register-stationary weights
 * and every fp32 operand split into two fp16 halves (22 of 24 mantissa bits) (3 x v_mfma_f32_16x16x32_f16 per
 * product sum, fp32 accumulation: the fp32 kernels' error against float64 at 3/16 of the matrix
 * cycles; no hidden round trip).  The CALLER guarantees the fp16 range: sqrt(C) max|norm_w| +
 * max|norm_b| and max_h(|W1_h|_1 * that + |b1_h|) below 6e4 (fused.FFN_F16_LIMIT; mssvt_amd/fused.py checks the
 * parameters once per version and keeps phases 3 otherwise).
 * Instantiated for (C,FF) in {(128,256),(64,128),(32,64)}; MSSVT_E_TOOLARGE otherwise.   */
int mssvt_ffn_fused(int n_rows, int C, int FF, const float *x_new, const float *x_in, const int *owner,
                    const float *norm_w, const float *norm_b, float eps, const float *W1,
                    const float *b1, const float *W2, const float *b2, float *y,
                    const float *next_norm_w, const float *next_norm_b, float next_eps, float *y_norm,
                    float *hidden, const int *num_rows_dev, int phases, void *stream);

/* W1 / W2 of mssvt_ffn_fused split into fp16 (hi, lo) MFMA fragments, once per parameter version: `packed` of
 * mssvt_ffn_packed_bytes(C, FF) bytes (0: shape not instantiated), passed as `hidden` with phases 4.          */
long long mssvt_ffn_packed_bytes(int C, int FF);
int mssvt_ffn_pack_weights(int C, int FF, const float *W1, const float *W2, void *packed, void *stream);

/* Table form of mssvt_block_interp_scatter: for every voxel owned by a list slot, tab_row (N,4)
 * int32 = the three rows of `attn` (row = w*nq + slot; empty slots / zero weights -> zero_row) and
 * tab_w (N,4) f32 = their inverse-distance weights.  The caller pre-fills tab_row with -1
 * (= voxel owned by no slot).                                                            */
int mssvt_block_interp_table(int nq, int n_upd, int use_interpolation, const int *indices,
                             const int *win_ind, const int *num_wins_dev, int win_capacity,
                             const int *win_vstart, const int *q_ind, const int *upd_ind,
                             const int *owner, const float *host_voxel_size3,
                             const float *host_range_min3, int zero_row, int *tab_row, float *tab_w,
                             void *stream);
/* mssvt_block_interp_table for several (query list, interpolation mode) variants of one plan in
 * ONE launch (host arrays of num_sets <= 4 entries).                                          */
int mssvt_block_interp_table_multi(int num_sets, const int *host_nq, const int *host_n_upd,
                                   const int *host_interp, const int *indices, const int *win_ind,
                                   const int *num_wins_dev, int win_capacity, const int *win_vstart,
                                   const int *const *host_q_ind, const int *const *host_upd_ind,
                                   const int *const *host_owner, const float *host_voxel_size3,
                                   const float *host_range_min3, const int *host_zero_row,
                                   int *const *host_tab_row, float *const *host_tab_w, void *stream);
/* mssvt_ffn_fused fed by that table: x = tab_row[v][0] < 0 ? 2*x_in[v]
 *                                      : x_in[v] + sum_i tab_w[v][i] * attn[tab_row[v][i]].
 * y may be NULL when phases == 4 and y_norm is given (round 6): only the next LayerNorm's output is stored -- the last
 * Block in front of a CompressBlock, which has no input residual (ref mssvt_backbone.py:370-385).                    */
int mssvt_ffn_fused_interp(int n_rows, int C, int FF, const float *x_in, const int *tab_row,
                           const float *tab_w, const float *attn, const float *norm_w,
                           const float *norm_b, float eps, const float *W1, const float *b1,
                           const float *W2, const float *b2, float *y, const float *next_norm_w,
                           const float *next_norm_b, float next_eps, float *y_norm, float *hidden,
                           const int *num_rows_dev, int phases, void *stream);

/* ======================================================================== *
 * Part 3 -- voxelizer front-end: the index part of
 *           DynamicVFE.forward (ref pcdet/models/backbones_3d/vfe/dynamic_vfe.py:83-93,
 *           114-118) without a sort: occupancy bitmap + popcount rank.
 * ======================================================================== */

/* int32 words of workspace for a (batch_size, X, Y, Z) grid. */
long long mssvt_voxelize_workspace_ints(int batch_size, int X, int Y, int Z);

/* points (P, point_stride) f32 rows [b, x, y, z, ...] -> voxel_coords (capacity,4) int32 rows
 * [b,z,y,x], sorted by (b,x,y,z) exactly like torch.unique of the reference's linear key;
 * point_voxel (P) int32 (nullable) = index of the point's voxel, -1 if outside the grid;
 * num_voxels_dev: device int receiving the voxel count.  host_*3: HOST float[3].          */
int mssvt_voxelize(const float *points, int point_stride, long long num_points, int batch_size,
                   const float *host_range_min3, const float *host_voxel_size3, int X, int Y, int Z,
                   int voxel_capacity, int *voxel_coords, int *point_voxel, int *num_voxels_dev,
                   int *workspace, void *stream);

/* Per-voxel reductions of DynamicVFE without torch_scatter (ref dynamic_vfe.py:98,111,128-129);
 * point_voxel (P) = unq_inv of mssvt_voxelize (-1: point outside the grid).
 * mssvt_voxel_mean_xyz: mean3 (N,3) f32 = scatter_mean of the x,y,z columns (points rows
 *   [b,x,y,z,...]); count (N) points per voxel; scratch_sum3: N*3 64-bit words (fixed-point
 *   sums, 2^-20 m: order independent -> deterministic).
 * mssvt_voxel_max: out (N,F) f32 = scatter_max of features (P,F) (voxels without a point: -inf). */
int mssvt_voxel_mean_xyz(const float *points, int point_stride, long long num_points,
                         const int *point_voxel, int num_voxels, float *mean3, int *count,
                         long long *scratch_sum3, void *stream);
int mssvt_voxel_max(const float *features, int F, long long num_points, const int *point_voxel,
                    int num_voxels, float *out, void *stream);

/* ======================================================================== *
 * Part 4 -- dense output: SparseTensor.dense() + the view of
 *           HeightCompression (ref mssvt_utils.py:6-19,50-62; height_compression.py:41-45) in one pass.
 * out (B, C, Z, Y, X) f32 (= (B, C*Z, Y, X) for the BEV backbone), every element written (zeros for empty
 * cells): out[b, c, z, y, x] = features[row(b,x,y,z), c] where the row comes from the set's hash table
 * map_table (B,H,2) (key x*Y*Z + y*Z + z -> row within the sample; v_bs_cnt (B) rows per sample).
 * ======================================================================== */
int mssvt_dense_bev(const float *features, int C, const int *map_table, int hash_size, const int *v_bs_cnt,
                    int batch_size, int x_max, int y_max, int z_max, float *out, void *stream);

/* Deterministic segmented row sum -- replaces the atomicAdd accumulation of the reference's gather backward passes
 * (ref: group_features_grad_kernel_stack, pcdet/ops/mssvt/src/group_features_gpu.cu:15-47;
 * gather_points_grad_kernel_fast, pcdet/ops/pointnet2/pointnet2_batch/src/sampling_gpu.cu:53-90;
 * group_points_grad_kernel_fast, .../group_points_gpu.cu:14-50), whose sum order changes run to run:
 *   dst[d][0..C) = sum over e in [csr_off[d], csr_off[d+1]) of (csr_w ? csr_w[e] : 1) * src[csr_idx[e]][0..C),
 * added in ascending e (an inverted index of the gather, built once per index set), so gradients are bit-identical
 * from run to run.  csr_off (n_dst+1) int32, csr_idx (nnz) int32 rows of src, csr_w (nnz) f32 or NULL,
 * src (R,C) f32, dst (n_dst,C) f32 (every row written; C % 4 == 0).  Also the forward of a weighted row gather. */
int mssvt_segment_sum_rows(int C, int n_dst, const int *csr_off, const int *csr_idx, const float *csr_w,
                           const float *src, float *dst, void *stream);

/* The same sum over explicit entry ranges [seg_start[d], seg_end[d]) (an empty range writes a zero row): lets the
 * caller cut a destination with thousands of contributions -- e.g. voxel 0 of a sample, the target of every FPS-picked
 * empty slot (ref mssvt_backbone.py:253-256) -- into fixed chunks summed by separate lane groups, and add the chunk
 * sums in chunk order with a second call: the order stays fixed, no lane group runs for the whole launch.       */
int mssvt_segment_sum_rows_ranges(int C, int n_dst, const int *seg_start, const int *seg_end, const int *csr_idx,
                                  const float *csr_w, const float *src, float *dst, void *stream);

/* The same sum on C columns of wider rows: src / dst row strides in floats (multiples of 4, >= C; both pointers at the
 * first of the C columns), and dst either written (accumulate 0) or added to (1: dst[d] = dst[d] + sum, one add per
 * element).  The gradient of a gather of a column range -- a head group of a Block, ref mssvt_backbone.py:260-268 --
 * lands in that range of the full-width gradient without a zero-filled temporary and an add.                    */
int mssvt_segment_sum_rows_strided(int C, int n_dst, const int *seg_start, const int *seg_end, const int *csr_idx,
                                   const float *csr_w, const float *src, int src_stride, float *dst, int dst_stride,
                                   int accumulate, void *stream);

/* The same sum with a residual epilogue: dst[d] = row_a[d] * res[d] + row_b[d] * sum (res, dst (n_dst,C); row_a, row_b
 * (n_dst) f32).  The tail of a Block under autograd in one launch -- 3-NN interpolation of the attention rows, the
 * "untouched voxels keep x_in" select, DropPath and the residual add (ref mssvt_backbone.py:298-340): row_b = the
 * row's DropPath factor, row_a = 1 for a voxel the attention updates, 1 + row_b for one it does not.               */
int mssvt_segment_sum_rows_residual(int C, int n_dst, const int *seg_start, const int *seg_end, const int *csr_idx,
                                    const float *csr_w, const float *src, const float *res, const float *row_a,
                                    const float *row_b, float *dst, void *stream);

/* Tokens of a Block's attention on compact rows (training path):
 *   tok[r][c] = (src ? src[rows[r]][c0+c] : 0) + relu(b[c0+c] + sum_{j<6} W6[c0+c][j] geo8[r][j]),  r < M, c < cg
 * = gathered feature + positional embedding of (offset to the window centre, window centre) (ref pos_proj,
 * mssvt_backbone.py:43-47, token sums :270-285; padded gathers group_features_gpu.cu:52-84, group_points_gpu.cu:56-91).
 * src (N,C) f32 or NULL, rows (M) int32, geo8 (M,8) f32 (columns 6, 7 unused), W6 (Ctot,6) / b (Ctot) = the Conv1d(6,C,1)
 * parameters, tok (M,cg) f32; cg in {16,32,64,128,256}, c0 % 4 == 0.
 * Backward of the embedding: _partial writes the slab of one token set (mssvt_train_tok_slab_floats(M,cg) floats) from
 * dtok (M,cg), the ReLU mask recomputed from geo8; _reduce adds the slabs of up to 4 token sets (host arrays: rows M,
 * first channel c0, width cg, slab address per set; sets with M <= 0 are skipped) in a fixed order into dW6 (C,6) and
 * db (C) (channels no set covers get 0).  The gradient of src is mssvt_segment_sum_rows_strided over the inverted
 * index of `rows`.  No atomics: bit-identical run to run.                                                        */
int mssvt_train_tok_forward(int M, int C, int c0, int cg, const int *rows, const float *src, const float *geo8,
                            const float *W6, const float *b, float *tok, void *stream);
long long mssvt_train_tok_slab_floats(int M, int cg);
int mssvt_train_tok_backward_partial(int M, int c0, int cg, const float *geo8, const float *W6, const float *b,
                                     const float *dtok, float *slab, void *stream);
int mssvt_train_tok_backward_reduce(int C, int num_sets, const int *host_M, const int *host_c0, const int *host_cg,
                                    const float *const *host_slabs, float *dW, float *db, void *stream);

/* Compact key sets of a window plan (index work of the training path; the padded key lists of ref
 * mssvt_backbone.py:253-268 without their empty slots).  kmeta (cap,K,4) f32 = per key slot (offset to the window centre
 * xyz, voxel row as int bits, < 0: empty), as mssvt_window_plan_two leaves it; K <= 64.
 * _counts: cnt[w] = valid slots of window w (0 for w >= *num_wins_dev), *total_dev += their sum (zero it first).
 * _compact: with off = the exclusive prefix sum of cnt, slot k of window w goes to position off[w] + (valid slots before
 * k): k_rows (voxel row), k_win (window), k_geo8 (8 f32: offset xyz, centre xyz from wcentre (cap,4), 0, 0 -- the
 * positional embedding's inputs for mssvt_train_tok_forward).                                                    */
int mssvt_train_key_counts(int cap, int K, const int *num_wins_dev, const float *kmeta, int *cnt, int *total_dev,
                           void *stream);
int mssvt_train_key_compact(int num_wins, int K, const float *kmeta, const float *wcentre, const int *off, int *k_rows,
                            int *k_win, float *k_geo8, void *stream);

/* The table of mssvt_block_interp_table (tab_row4 / tab_w4 (N,4): three rows of the padded attention buffer + weights per
 * voxel, row < 0 in .x: a voxel the attention does not update) in compact form for the training path: idx3 / w3 (3 per
 * voxel) name compact attention rows through inv (padded row -> compact row); an untouched voxel gets row R (the zero
 * row) with weight 0; owned (N) bytes = 1 where the attention updates the voxel (ref mssvt_backbone.py:298-338).   */
int mssvt_train_interp_compact(int N, int R, const int *inv, const int *tab_row4, const float *tab_w4, int *idx3, float *w3,
                               unsigned char *owned, void *stream);

/* Compact (window, voxel) pairs of a CompressBlock's window lists (training path; index work): k_ind (num_wins,ns) int32 =
 * positions inside the window's run of voxel rows, < 0: empty (mssvt_window_plan_one), ns <= 64.  _counts: cnt[w] = valid
 * slots, *total_dev += their sum (zero it first).  _compact, with off = the exclusive prefix sum of cnt: pair_vox (voxel
 * row win_vstart[w] + k), pair_win, geo8 (8 f32: voxel centre - window centre, window centre, 0, 0; centres as
 * (index + 0.5) * cell + min in three rounded fp32 operations, ref with_coords mssvt_backbone.py:132-137) at the prefix
 * position of the slot.  indices (N,4) / win_ind (num_wins,4) int32 (b,z,y,x).                                    */
int mssvt_train_list_counts(int num_wins, int ns, const int *k_ind, int *cnt, int *total_dev, void *stream);
int mssvt_train_pairs_compact(int num_wins, int ns, const int *k_ind, const int *win_vstart, const int *off,
                              const int *indices, const int *win_ind, const float *host_voxel_size3,
                              const float *host_range_min3, const float *host_win_size3, int *pair_vox, int *pair_win,
                              float *geo8, void *stream);

/* Compact query rows of a query pattern of a window plan (training path; index work): from what mssvt_plan_order leaves
 * (row_meta (rows,4) f32 = offset to the window centre xyz + voxel row as int bits; row_src (rows,2) int32 = window,
 * padded attention row) the voxel rows q_rows (R) and the positional embedding's inputs q_geo8 (R,8: offset, centre from
 * wcentre (cap,4), 0, 0) of the first R rows.                                                                    */
int mssvt_train_query_sets(int R, const float *row_meta, const int *row_src, const float *wcentre, int *q_rows,
                           float *q_geo8, void *stream);

/* Inverse of a gather of DISTINCT rows (dst[i] = src[idx[i]], every source row named at most once: a window's query rows
 * when every window size is odd, the voxels of disjoint windows) as ranges for mssvt_segment_sum_rows_ranges over the
 * n_src source rows: seg_start[v] = v, seg_end[v] = bwd_end[v] (v + 1 or v), csr_idx = bwd_idx.  inv_scratch: n_src ints.
 * Replaces the sort of mssvt_csr_transpose for such gathers (the reference's atomicAdd backward needs no order there). */
int mssvt_train_unique_inverse(int nnz, int n_src, const int *idx, int *inv_scratch, int *bwd_idx, int *bwd_end,
                               void *stream);

/* ========================================================================
 * Post-processing behind the backbone: rotated BEV NMS of CenterHead's boxes.
 * ref: iou3d_nms_cuda.nms_gpu, pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:90-135 (host loop) +
 * iou3d_nms_kernel.cu:107-278 (box_overlap / iou_bev / nms_kernel).  boxes_sorted (N,7) f32
 * [x,y,z,dx,dy,dz,heading] in descending score order (the caller sorts, as the reference's Python does,
 * iou3d_nms_utils.py:83-98); a box suppresses every later box whose BEV IoU exceeds thresh (strict).
 * keep (N) int32 receives the kept positions (ascending), *num_keep_dev their count -- the greedy walk runs
 * on the device, nothing but these is read back.  workspace: mssvt_nms_workspace_bytes(N) bytes. N <= 16384.
 * ======================================================================== */
long long mssvt_nms_workspace_bytes(int num_boxes);
int mssvt_nms_bev(int num_boxes, const float *boxes_sorted, float thresh, void *workspace, int *keep,
                  int *num_keep_dev, void *stream);

/* Weight / bias gradient of an nn.Linear over compact rows (training path; what autograd's library GEMM computes for
 * the reference's to_qs / to_kvs / projs / linear1 / linear2, ref mssvt_utils.py:80-83, mssvt_backbone.py:25-27):
 *   dW (Cout,Cin) = dY^T X,  db (Cout) = column sums of dY (db may be NULL);  X (M,Cin), dY (M,Cout) f32 row-major.
 * Deterministic split over the M rows on the fp32 matrix cores: per-slice partial slabs in `workspace`
 * (mssvt_linear_wgrad_workspace_floats(M,Cin,Cout) floats), added in slice order.  Cin, Cout multiples of 4,
 * ceil(Cout/16) * (ceil(Cin/16) + 1) <= 160 (256 x 128, 128 x 256 ...); MSSVT_E_TOOLARGE otherwise.           */
long long mssvt_linear_wgrad_workspace_floats(int M, int Cin, int Cout);
int mssvt_linear_wgrad(int M, int Cin, int Cout, const float *X, const float *dY, float *dW, float *db,
                       float *workspace, void *stream);

/* Y = X B^T + bias over compact rows on the fp32 matrix cores (training path), for the SMALL weight matrices of a Block
 * (64 x 64 to_qs / projs of a head group: ref mssvt_utils.py:80-83, mssvt_backbone.py:25-27) and, with transpose_w, their
 * input gradient dX = dY W -- library GEMMs under autograd in the reference, which pick poor tiles on these shapes.
 *   Y[m][n] = out_scale * act(bias[n] + sum_k X[m][k] B[n][k]),  B = W (N,K) row-major, or B[n][k] = W[k][n] with W (K,N)
 * (transpose_w != 0); out_scale = 1 or the attention scale of a query projection (ref mssvt_utils.py:131-133);
 * X (M,K) f32 with row stride ldx (floats, % 4 == 0), Y (M,N) with row stride ldy, bias (N) or NULL, relu != 0 clamps Y
 * at 0.  K, N in {64,128} (mssvt_linear_rows_supported); MSSVT_E_TOOLARGE otherwise.                                  */
int mssvt_linear_rows_supported(int K, int N);
int mssvt_linear_rows(int M, int K, int N, const float *X, int ldx, const float *W, int transpose_w, const float *bias,
                      int relu, float out_scale, float *Y, int ldy, void *stream);

/* Window attention on compact rows, forward and backward (training path; the softmax(QK^T)V of ref
 * mssvt_utils.py:131-149 for one head group, without the padded (windows, slots) layout and the -100 mask):
 * window w owns query rows [q_off[w], +q_cnt[w]) of q (R,cg) (already scaled) and key rows [k_off[w], +k_cnt[w]) of
 * kv (Kn,2cg) = [K|V]; ranges of different windows are disjoint.  fwd: O (R,cg), lse (R,heads) = log-sum-exp of the
 * scores.  bwd: dq (R,cg), dkv (Kn,2cg) from dO; every row is written by exactly one wave in a fixed order (no
 * atomics, bit-identical run to run).  cg = heads * hd <= 128, hd in {4,8,16,32,64}; windows without keys give O = 0. */
int mssvt_pair_attention_fwd(int nw, int cg, int heads, int hd, const int *q_off, const int *q_cnt, const int *k_off,
                             const int *k_cnt, const float *q, const float *kv, float *O, float *lse, void *stream);
int mssvt_pair_attention_bwd(int nw, int cg, int heads, int hd, const int *q_off, const int *q_cnt, const int *k_off,
                             const int *k_cnt, const float *q, const float *kv, const float *O, const float *lse,
                             const float *dO, float *dq, float *dkv, void *stream);

/* Inverted index of a (weighted) row gather dst[d] = sum_{e in [off[d],off[d+1])} w[e] src[idx[e]] (off NULL: one entry
 * per row), built on the device: grad_src[s] = sum_{p in [t_off[s],t_off[s+1])} t_w[p] grad_dst[t_idx[p]] with the
 * entries of a source row in ascending e -- the fixed summation order that replaces the reference's atomicAdd backward
 * (ref group_features_gpu.cu:15-47, sampling_gpu.cu:53-90).  Entries of drop_src (or out of [0,n_src)) are left out.
 * *max_count = the longest list if one is longer than long_list, else 0.  t_off (n_src+1), t_idx / t_w (nnz; t_w NULL
 * iff w NULL); workspace of mssvt_csr_transpose_workspace_bytes(nnz, n_src) bytes.                                    */
long long mssvt_csr_transpose_workspace_bytes(int nnz, int n_src);
int mssvt_csr_transpose(int nnz, int n_dst, int n_src, const int *off, const int *idx, const float *w, int drop_src,
                        int long_list, int *t_off, int *t_idx, float *t_w, int *max_count, void *workspace, void *stream);

/* ======================================================================== *
 * Part 5 -- a whole backbone forward behind ONE call (round 4).
 *
 * ref: MixedScaleSparseTransformer.forward, pcdet/models/backbones_3d/mssvt_backbone.py:450-472, which drives its blocks
 * from Python (>= 5B + L(5B+2) host syncs, ~100 launches per Block).  A frame object describes one resolution level
 * of the network -- L two-scale Blocks that share one window configuration, closed by a CompressBlock over pillar
 * windows [1,1,z] -- and mssvt_frame_forward enqueues the entry points of part 2 for it in the order the Python
 * module path (mssvt_amd/fused.py) issues them, out of ONE caller-owned workspace: same kernels, same arguments,
 * bit-identical results, no allocation and no host synchronisation inside the call.  The object holds parameter
 * POINTERS (device memory owned by the caller: rebuild it when a parameter moves), one pinned 4-KiB host buffer and
 * one event for the frame's single device-to-host hand-over.  add_* return MSSVT_E_TOOLARGE for shapes this path
 * does not cover; the caller keeps its own path for those.
 * Input: a voxel list sorted by (b,x,y,z) (what DynamicVFE emits); any other order is reported in the status
 * word (MSSVT_ST_UNSORTED) and the outputs are then empty.
 * ======================================================================== */
#define MSSVT_ST_UNSORTED 8 /* level status: the voxel list is not strictly ascending in (b,x,y,z) */

int mssvt_frame_create(void **frame_out);
int mssvt_frame_destroy(void *frame);
/* Level geometry (ref mssvt_backbone.py:436-448): grid [x,y,z] (z <= 64), voxel size, point cloud range
 * [x0,y0,z0,x1,y1,z1] (HOST arrays), channel width C of every block and FF of every feed-forward layer ((C,FF)
 * instantiated in csrc/ffn.hip).  Clears the block list.                                                        */
int mssvt_frame_set_level(void *frame, int batch_size, int x_max, int y_max, int z_max, int hash_size,
                          const float *host_voxel_size3, const float *host_range6, int C, int FF);
/* One MixedScaleSparseTransformerBlock (ref mssvt_backbone.py:11-347): window configuration (arguments of
 * mssvt_window_plan_two: win1 size, list capacities, the four offset tables on the device, their footprint and packed
 * form, key_num_sample, max_num_wins -- identical for every Block of the level), cbs_pattern (0 even / 1 odd / 2 win1
 * queries), use_feature_interpolation, norm1, the attention parameters (arguments of mssvt_block_attention; two head
 * groups), attn_mode 0 = mssvt_block_attention, 1 = mssvt_block_attention_kv16 (host_packed: its blobs or NULL),
 * 2 = mssvt_block_attention_bf16; norm2 + linear1 / linear2 and the fragments of mssvt_ffn_pack_weights.           */
int mssvt_frame_add_block(
    void *frame, const int *host_win1_size3, int max_num_odd, int max_num_even, int max_num_win1, int max_num_win2,
    int num_odd, int num_even, int num_win1, int num_win2, const int *vox_query_odd, const int *vox_query_even,
    const int *vox_query_win1, const int *vox_query_win2, const int *host_footprint4, const int *packed_offsets,
    int key_num_sample, int max_num_wins, int cbs_pattern, int use_interpolation, const float *norm1_w,
    const float *norm1_b, float norm1_eps, int num_groups, const int *host_c0, const int *host_cg, const int *host_heads,
    int head_dim, float scale, const float *const *host_Wq, const float *const *host_bq, const float *const *host_Wkv,
    const float *const *host_bkv, const float *const *host_Wo, const float *const *host_bo,
    const void *const *host_packed, const float *Wpos, const float *bpos, int attn_mode, const float *norm2_w,
    const float *norm2_b, float norm2_eps, const float *W1, const float *b1, const float *W2, const float *b2,
    const void *ffn_packed);
/* The MixedScaleSparseTransformerCompressBlock that ends the level (ref mssvt_backbone.py:349-398): pillar windows
 * [1,1,z] whose offset table stays inside the window (one lane per window in the plan kernel), one head group;
 * arguments of mssvt_window_plan_one / mssvt_compress_fused / mssvt_ffn_fused.  compress_ws_packed: the fragments of
 * mssvt_compress_ws_pack when the caller has checked mssvt_compress_ws's preconditions (the table lists every cell of
 * the slab, fp16 range) -- the attention is then that one launch -- or NULL.                                          */
int mssvt_frame_add_compress(
    void *frame, const int *host_win_size3, int max_num_win1, int num_win1, const int *vox_query_win1, int max_num_wins,
    const float *norm1_w, const float *norm1_b, float norm1_eps, const float *Wpos1, const float *bpos1, const float *Wpos2,
    const float *bpos2, const float *Wq, const float *bq, const float *Wkv, const float *bkv, const float *Wo, const float *bo,
    int head_dim, float scale, int split_f16, const float *norm2_w, const float *norm2_b, float norm2_eps, const float *W1,
    const float *b1, const float *W2, const float *b2, const void *ffn_packed, const void *compress_ws_packed);
/* on != 0: the first norm1 and the CompressBlock's pillar plan run on a stream of the frame object, under the Blocks'
 * plan kernel (two event edges per frame; results unchanged).                                                     */
int mssvt_frame_set_overlap(void *frame, int on);
/* Bytes of workspace a forward over num_voxels voxels needs (0: frame incomplete). */
long long mssvt_frame_workspace_bytes(void *frame, int num_voxels);
/* Enqueue the forward on `stream`.  features (N,C) f32, indices (N,4) int32 [b,z,y,x]; workspace: 256-byte aligned,
 * contents irrelevant (reusable by the next frame on the same stream).  Outputs, capacity-sized and caller-owned:
 * out_features (N,C) / out_indices (N,4): the first `rows` rows are the output voxel set (window order);
 * out_table (B,H,2): its hash table (filled here, -1 prefill included); out_counts (B) rows per sample.
 * `rows` and the status words travel to the host through the frame's pinned buffer: mssvt_frame_wait_words blocks
 * until they have landed (long before the frame's kernels end) and copies words [0, num_words) out:
 * [0] level status (MSSVT_ST_UNSORTED), [64] status / [65] window count of the Blocks' partition,
 * [128] status / [129] window count of the CompressBlock's partition = rows.                                     */
int mssvt_frame_forward(void *frame, int num_voxels, const float *features, const int *indices, void *workspace,
                        long long workspace_bytes, float *out_features, int *out_indices, int *out_table,
                        int *out_counts, void *stream);
int mssvt_frame_wait_words(void *frame, int *host_out, int num_words);

/* Y = out_scale * act(X B^T + bias) over compact rows with split-fp16 matrix operands (csrc/linear_rows_h.hip; the
 * training path's nn.Linear forward and, with transpose_w, its input gradient dX = dY W -- ref mssvt_backbone.py:339-343,
 * mssvt_utils.py:80-83 through autograd): B = W (N, K) row-major, or B[n][k] = W[k][n] with transpose_w (W then (K, N)).
 * (K, N) in {(128,256), (256,128), (64,128), (128,64), (128,128), (64,64)}; ldx >= K, ldy >= N, multiples of 4.  Rows of X
 * and the weight matrix are normalised by powers of two inside the kernel: no range precondition.                    */
int mssvt_linear_rows_h_supported(int K, int N);
int mssvt_linear_rows_h(int M, int K, int N, const float *X, int ldx, const float *W, int transpose_w, const float *bias,
                        int relu, float out_scale, float *Y, int ldy, void *stream);

/* DynamicVFE's two PFN layers in eval mode as two launches (csrc/pfn_fused.hip; ref
 * pcdet/models/backbones_3d/vfe/dynamic_vfe.py:96-131, PFNLayerV2 :14-52), default configuration: 5 point features + cluster
 * offset + voxel-centre offset (11 inputs), NUM_FILTERS [64, 128].  points (P, stride >= 6) f32 rows [b, x, y, z, f4, f5];
 * point_voxel (P) int32, -1 outside the grid; mean3 (N, 3) = mssvt_voxel_mean_xyz; voxel_coords (N, 4) int32 [b, z, y, x];
 * host_voxel_size3 / host_offset3: HOST float[3] (offset = voxel_size / 2 + range_min); layer parameters as the state dict holds
 * them (pfn.{0,1}.0.weight / .bias, pfn.{0,1}.1.weight / .bias / .running_mean / .running_var, eps); x1_scratch (P, 64),
 * m1_scratch (max(N,1), 64), x2_scratch (P, 128): caller-owned; out (N, 128) = the voxel features.                                       */
int mssvt_pfn_fused_64_128(const float *points, int point_stride, long long num_points, const int *point_voxel,
                           int num_voxels, const float *mean3, const int *voxel_coords, const float *host_voxel_size3,
                           const float *host_offset3, const float *W1, const float *b1, const float *bn1_w,
                           const float *bn1_b, const float *bn1_mean, const float *bn1_var, float bn1_eps, const float *W2,
                           const float *b2, const float *bn2_w, const float *bn2_b, const float *bn2_mean,
                           const float *bn2_var, float bn2_eps, float *x1_scratch, float *m1_scratch, float *x2_scratch,
                           float *out, void *stream);

/* Round 6: the same two layers AND the cluster-centre mean over points grouped by voxel (csrc/pfn_sorted.hip): a counting
 * sort on point_voxel, then every reduction is a loop over a voxel's run of rows -- no atomics on feature rows, no -inf
 * fills, the per-point layer-2 output never reaches memory.  Results bit-identical to mssvt_voxel_mean_xyz +
 * mssvt_pfn_fused_64_128.  workspace: mssvt_pfn_sorted_workspace_ints(P, N) int32; x1_scratch (P, 64), m1_scratch (N, 64):
 * caller-owned; out (N, 128).  Every voxel must hold at least one point (mssvt_voxelize's output does).                  */
long long mssvt_pfn_sorted_workspace_ints(long long num_points, int num_voxels);
int mssvt_pfn_sorted_64_128(const float *points, int point_stride, long long num_points, const int *point_voxel,
                            int num_voxels, const int *voxel_coords, const float *host_voxel_size3,
                            const float *host_offset3, const float *W1, const float *b1, const float *bn1_w,
                            const float *bn1_b, const float *bn1_mean, const float *bn1_var, float bn1_eps, const float *W2,
                            const float *b2, const float *bn2_w, const float *bn2_b, const float *bn2_mean,
                            const float *bn2_var, float bn2_eps, int *workspace, float *x1_scratch, float *m1_scratch,
                            float *out, void *stream);

/* ======================================================================== *
 * Part 6 -- timing-only launches (csrc/ceiling.hip; no reference counterpart, nothing reads their output): the byte and
 *           instruction mix of k_ffn_ws / k_attn_kvh on the frame's real tables with every dependency between the
 *           phases removed -- the ceiling of the structure, reported by bench.py as roofline.ceiling_us.
 * ======================================================================== */
int mssvt_ceiling_ffn_ws(int n_rows, const float *x_in, const int *tab_row, const float *tab_w, const float *attn,
                         const void *fragments_256k, float *y, float *y_norm, void *stream);
/* round 6: the same launch with the mix as a parameter -- variant = 100 * mode + filler index; mode 0 no matrix instructions,
 * 1 = 48 v_mfma_f32_16x16x32_f16 per wave and 16-row tile (the product's), 2 = 48 v_mfma_f32_32x32x16_f16 per wave and
 * 32-row tile (the same FLOP per row); filler index -> {0, 72, 144, 216, 288, 400} vector instructions per wave and tile. */
int mssvt_ceiling_ffn_mix(int variant, int n_rows, const float *x_in, const int *tab_row, const float *tab_w,
                          const float *attn, const void *fragments_256k, float *y, float *y_norm, void *stream);
int mssvt_ceiling_attn_kvh(int C, int c0_group0, int c0_group1, int K, const float *xhat, const float *kmeta0,
                           const float *kmeta1, const int *perm, const int *num_active_dev, const int *q_off,
                           const int *nq_valid, int row_capacity, int win_capacity, float *qbuf, void *stream);

/* round 6: variant 1 = the mix of "Wv applied at the end of the window launch" (24 more matrix + ~40 more vector
 * instructions per pass of 4 queries, the hand-off a quarter of the bytes); variant 0 = mssvt_ceiling_attn_kvh.          */
int mssvt_ceiling_attn_kvh_variant(int variant, int C, int c0_group0, int c0_group1, int K, const float *xhat,
                                   const float *kmeta0, const float *kmeta1, const int *perm, const int *num_active_dev,
                                   const int *q_off, const int *nq_valid, int row_capacity, int win_capacity, float *qbuf,
                                   void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MSSVT_HIP_H */
