// frame.hip -- one C entry point that enqueues a whole backbone forward (round 4).
//
// The reference's MixedScaleSparseTransformer.forward (ref: pcdet/models/backbones_3d/mssvt_backbone.py:450-472) drives its
// blocks from Python: ~100 launches and >= 5B + 2 host syncs per Block.  mssvt_amd/fused.py cut that to ~34 launches and no
// sync, but the ~65 allocations, ~15 ctypes calls and the bookkeeping around them still cost ~630 us of interpreter time
// per frame -- as much as the kernels.  Here the same sequence of the SAME entry points (same arguments, same order:
// results are bit-identical to the Python-driven fused path) is issued from C++ out of one caller-owned workspace:
//
//     fill (-1 arena, zero region, the output table) + the first norm1 in the same launch    k_frame_fill_ln
//     level set-up of the (b,x,y,z)-sorted input list + early device-to-host copy            mssvt_level_setup_sorted_pillars
//     window plan of the two-scale Blocks (+ interpolation tables), work orders              mssvt_window_plan_two, mssvt_plan_order_multi
//     per Block: window attention, FFN tail (emits the next block's norm1)                   mssvt_block_attention*, mssvt_ffn_fused_interp
//     CompressBlock: attention (its pillar lists come out of the level set-up), FFN tail      mssvt_compress_fused, mssvt_ffn_fused
//
// A frame object (mssvt_frame_create) only holds the description of the network: window configuration, parameter
// pointers (device memory owned by the caller), one pinned 4-KiB host buffer and one event for the frame's single
// device-to-host hand-over (status words + the data-dependent output row count).  It launches on the caller's stream,
// allocates nothing per frame and keeps no device state between frames.  Shapes it does not cover return
// MSSVT_E_TOOLARGE from the add_* calls: the caller keeps its own path for those (mssvt_amd/fused.py).
#include <new>
#include <vector>

#include "common.hip.h"
#include "../../include/mssvt_hip.h"

namespace {

struct FrBlock {
    int cbs_pattern, interp;
    const float *n1w, *n1b;
    float n1eps;
    int ng, c0[2], cg[2], heads[2], head_dim;
    float scale;
    const float *Wq[2], *bq[2], *Wkv[2], *bkv[2], *Wo[2], *bo[2];
    const void *packed[2];
    bool have_packed;
    const float *Wp, *bp;
    int attn_mode;  // 0: fp32 matrix instruction, 1: split-fp16 operands (kv16), 2: bf16 operands
    const float *n2w, *n2b;
    float n2eps;
    const float *W1, *b1, *W2, *b2;
    const void *ffn_packed;
};

struct FrPlanCfg {
    int ws[3], n_o, n_e, n1, n2, num_o, num_e, num_1, num_2, K, max_wins;
    const int *t_o, *t_e, *t_1, *t_2, *packed_offsets;
    int fp4[4];
};

struct FrCompress {
    int ws[3], ns, num_1, max_wins;
    const int *t_1;
    const float *n1w, *n1b;
    float n1eps;
    const float *Wp1, *bp1, *Wp2, *bp2, *Wq, *bq, *Wkv, *bkv, *Wo, *bo;
    int head_dim;
    float scale;
    int split_f16;
    const float *n2w, *n2b;
    float n2eps;
    const float *W1, *b1, *W2, *b2;
    const void *ffn_packed;
    const void *ws_packed;  // fragments of mssvt_compress_ws_pack, or null: mssvt_compress_fused
};

struct Frame {
    bool level_set = false, has_cmp = false;
    int B = 0, X = 0, Y = 0, Z = 0, H = 0, C = 0, FF = 0;
    float vs[3], range[6];
    FrPlanCfg plan;
    std::vector<FrBlock> blocks;
    FrCompress cmp;
    hipEvent_t ready = nullptr;
    int *host_words = nullptr;  // pinned
    int words = 0;
    // optional: the launch that neither needs nor feeds the index chain (first norm1)
    // on a second stream, under the Blocks' plan kernel (VALU / latency bound; the LayerNorm is HBM bound)
    int overlap = 0;
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};

constexpr int FR_HOST_WORDS = 1024;

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Bump {
    char *base;
    size_t off, cap;
    template <typename T>
    T *take(size_t count) {
        off = al256(off);
        T *p = reinterpret_cast<T *>(base + off);
        off += count * sizeof(T);
        return p;
    }
};

// a[0..n_a) = -1, b[0..n_b) = -1, c[0..n_c) = 0, d[0..n_d) = 0 (counts in ints; 16-byte aligned bases)
__global__ void __launch_bounds__(256) k_frame_fill(int *a, long long n_a, int *b, long long n_b, int *c, long long n_c, int *d,
                                                   long long n_d) {
    const long long qa = n_a >> 2, qb = n_b >> 2, qc = n_c >> 2, qd = n_d >> 2, stride = (long long)gridDim.x * blockDim.x;
    const int4 neg = make_int4(-1, -1, -1, -1), zero = make_int4(0, 0, 0, 0);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < qa + qb + qc + qd; i += stride) {
        if (i < qa) reinterpret_cast<int4 *>(a)[i] = neg;
        else if (i < qa + qb) reinterpret_cast<int4 *>(b)[i - qa] = neg;
        else if (i < qa + qb + qc) reinterpret_cast<int4 *>(c)[i - qa - qb] = zero;
        else reinterpret_cast<int4 *>(d)[i - qa - qb - qc] = zero;
    }
    if (blockIdx.x == 0 && threadIdx.x < 16) {  // the (at most 3) ints behind the last full quad of each region
        const int reg = threadIdx.x >> 2, t = threadIdx.x & 3;
        int *base = reg == 0 ? a : reg == 1 ? b : reg == 2 ? c : d;
        const long long nn = reg == 0 ? n_a : reg == 1 ? n_b : reg == 2 ? n_c : n_d, done = (nn >> 2) << 2;
        if (done + t < nn) base[done + t] = reg < 2 ? -1 : 0;
    }
}

// The same fill with the frame's first LayerNorm riding along: workgroups [0, fill_blocks) fill, the others normalise 4 * (64 /
// LPR) rows each -- two independent pieces of work at the very front of the frame in ONE launch (each launch of the frame
// has a floor of ~4.5 us in the profiler's accounting; the fill alone is 5.6 us for 8.6 MB).
template <int LPR>
__global__ void __launch_bounds__(256) k_frame_fill_ln(int *a, long long n_a, int *b, long long n_b, int *c, long long n_c, int *d,
                                                      long long n_d, int fill_blocks, const float *x, int n, const float *w,
                                                      const float *bias, float eps, float *y) {
    if ((int)blockIdx.x >= fill_blocks) {
        layer_norm_rows<LPR>(x, n, w, bias, eps, y, blockIdx.x - fill_blocks);
        return;
    }
    const long long qa = n_a >> 2, qb = n_b >> 2, qc = n_c >> 2, qd = n_d >> 2, stride = (long long)fill_blocks * blockDim.x;
    const int4 neg = make_int4(-1, -1, -1, -1), zero = make_int4(0, 0, 0, 0);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < qa + qb + qc + qd; i += stride) {
        if (i < qa) reinterpret_cast<int4 *>(a)[i] = neg;
        else if (i < qa + qb) reinterpret_cast<int4 *>(b)[i - qa] = neg;
        else if (i < qa + qb + qc) reinterpret_cast<int4 *>(c)[i - qa - qb] = zero;
        else reinterpret_cast<int4 *>(d)[i - qa - qb - qc] = zero;
    }
    if (blockIdx.x == 0 && threadIdx.x < 16) {
        const int reg = threadIdx.x >> 2, t = threadIdx.x & 3;
        int *base = reg == 0 ? a : reg == 1 ? b : reg == 2 ? c : d;
        const long long nn = reg == 0 ? n_a : reg == 1 ? n_b : reg == 2 ? n_c : n_d, done = (nn >> 2) << 2;
        if (done + t < nn) base[done + t] = reg < 2 ? -1 : 0;
    }
}

inline size_t ints_al(size_t n) { return (n + 63) / 64 * 64; }  // 256-byte pieces, multiples of 4 ints

int q_slots(const FrPlanCfg &p, int pattern) { return pattern == 0 ? p.n_e : pattern == 1 ? p.n_o : p.n1; }
int q_list(int pattern) { return pattern == 1 ? 0 : pattern == 0 ? 1 : 2; }  // cbs_pattern -> list: 0 odd, 1 even, 2 win1

// Everything the forward carves out of the workspace, computed in one place so that mssvt_frame_workspace_bytes and
// mssvt_frame_forward cannot disagree: run with base = nullptr to size, with the real base to place.
struct Layout {
    // -1 arena
    int *neg0;
    size_t neg_ints;
    int *tab_rows, *pair_win;
    // zero region
    int *zero0;
    size_t zero_ints;
    int *status, *hdr[2], *start;
    unsigned long long *occ;
    float *attn_zero;  // one zero row behind the attention buffers
    // the rest
    int *cnt, *vbase, *scratch, *win_blk, *vcount_blk;
    float *xhat0;
    int *win_vstart, *nq_valid;
    float *qmeta[3], *kmeta[2], *wcentre, *tab_w;
    int n_tabs, tab_pat[4], tab_interp[4];
    int n_pat, pats[3];
    int *perm[3], *n_act[3], *q_off[3], *row_src[3], *n_rows[3];
    float *row_meta[3];
    long long row_cap;
    float *qbuf, *attn[3];
    int attn_zero_row[3];
    float *x[2], *xh[2];
    int *c_k_ind, *c_win_vstart, *c_win_cnt, *c_pair_base, *c_pair_vox;
    float *c_qp, *c_ktok, *c_score, *c_vp, *c_new;
    size_t total;
};

int pattern_slot(const Layout &L, int pattern) {
    for (int i = 0; i < L.n_pat; ++i)
        if (L.pats[i] == pattern) return i;
    return -1;
}

void make_layout(const Frame &f, int n, char *base, Layout &L) {
    Bump b{base, 0, 0};
    const size_t N = (size_t)(n > 0 ? n : 1), cap = N;
    const int B = f.B, C = f.C;
    const FrPlanCfg &p = f.plan;
    // distinct query patterns / interpolation tables of the Blocks (order of first use, as fused._plan_tables)
    L.n_pat = 0;
    L.n_tabs = 0;
    for (const FrBlock &k : f.blocks) {
        if (pattern_slot(L, k.cbs_pattern) < 0) L.pats[L.n_pat++] = k.cbs_pattern;
        bool seen = false;
        for (int t = 0; t < L.n_tabs; ++t) seen = seen || (L.tab_pat[t] == k.cbs_pattern && L.tab_interp[t] == k.interp);
        if (!seen) {
            L.tab_pat[L.n_tabs] = k.cbs_pattern;
            L.tab_interp[L.n_tabs++] = k.interp;
        }
    }
    // ---- -1 arena
    L.neg0 = b.take<int>(0);
    L.tab_rows = b.take<int>(ints_al((size_t)L.n_tabs * N * 4));
    L.pair_win = b.take<int>(ints_al(cap));
    // (the K4 lists themselves only for the three-launch form; the level set-up writes the listed slots only)
    L.c_k_ind = f.has_cmp && !f.cmp.ws_packed ? b.take<int>(ints_al(cap * (size_t)f.cmp.ns)) : nullptr;
    b.off = al256(b.off);
    L.neg_ints = (size_t)(b.base + b.off - reinterpret_cast<char *>(L.neg0)) / 4;
    // work orders (fused._work_order / prepare_group): row capacity = max over the patterns of min(N * overlap, cap * nq)
    long long overlap = 1;
    for (int i = 0; i < 3; ++i) overlap *= p.ws[i] % 2 == 0 ? 2 : 1;
    L.row_cap = 1;
    for (int i = 0; i < L.n_pat; ++i) {
        long long rc = (long long)N * overlap, lim = (long long)cap * q_slots(p, L.pats[i]);
        if (rc > lim) rc = lim;
        if (rc > L.row_cap) L.row_cap = rc;
    }
    // ---- zero region (the block of fused._sorted_level: status | partition headers | sample starts | occupancy words)
    L.zero0 = b.take<int>(0);
    L.status = b.take<int>(64);
    L.hdr[0] = b.take<int>(64);
    L.hdr[1] = b.take<int>(64);
    L.start = b.take<int>(ints_al(B + 1));
    L.occ = reinterpret_cast<unsigned long long *>(b.take<int>(ints_al((size_t)2 * B * f.X * f.Y)));
    b.off = al256(b.off);
    L.zero_ints = (size_t)(b.base + b.off - reinterpret_cast<char *>(L.zero0)) / 4;
    // ---- uninitialised
    L.cnt = b.take<int>(B);
    L.vbase = b.take<int>((size_t)B * f.X * f.Y);
    L.scratch = b.take<int>((size_t)mssvt_level_sorted_scratch_ints(B, f.X, f.Y));
    L.win_blk = b.take<int>(N * 4);
    L.vcount_blk = b.take<int>(B);
    L.xhat0 = b.take<float>(N * C);
    // (the plan's list rows, key indices / masks and owner arrays are not asked for: its consumers here read the resolved
    // metadata and the interpolation tables)
    for (int g = 0; g < 2; ++g) L.kmeta[g] = b.take<float>(cap * p.K * 4);
    L.win_vstart = b.take<int>(cap);
    for (int l = 0; l < 3; ++l) L.qmeta[l] = nullptr;
    for (int i = 0; i < L.n_pat; ++i) {
        const int l = q_list(L.pats[i]);
        L.qmeta[l] = b.take<float>(cap * (size_t)q_slots(p, L.pats[i]) * 4);
    }
    L.wcentre = b.take<float>(cap * 4);
    L.nq_valid = b.take<int>(3 * cap);
    L.tab_w = b.take<float>((size_t)L.n_tabs * N * 4);
    for (int i = 0; i < L.n_pat; ++i) {
        L.perm[i] = b.take<int>(cap);
        L.n_act[i] = b.take<int>(1);
        L.q_off[i] = b.take<int>(cap);
        L.row_meta[i] = b.take<float>((size_t)L.row_cap * 4);
        L.row_src[i] = b.take<int>((size_t)L.row_cap * 2);
        L.n_rows[i] = b.take<int>(1);
    }
    // hand-off scratch of the attention launches: one row per valid query, one region per head group
    size_t qwidth = 0;
    if (!f.blocks.empty())
        for (int g = 0; g < f.blocks[0].ng; ++g) qwidth += (size_t)4 * ((f.blocks[0].heads[g] + 3) / 4) * f.blocks[0].cg[g];
    L.qbuf = b.take<float>((size_t)L.row_cap * qwidth);
    // attention rows (cap * nq per pattern) in ONE piece that ends in a zero row: buffer i starts at row offset off_i,
    // the shared zero row is row total - 1 - off_i of it
    size_t rows = 0;
    for (int i = 0; i < L.n_pat; ++i) rows += cap * (size_t)q_slots(p, L.pats[i]);
    float *big = b.take<float>((rows + 1) * C);
    size_t off_rows = 0;
    for (int i = 0; i < L.n_pat; ++i) {
        L.attn[i] = big + off_rows * C;
        L.attn_zero_row[i] = (int)(rows - off_rows);
        off_rows += cap * (size_t)q_slots(p, L.pats[i]);
    }
    L.attn_zero = big + rows * C;
    for (int i = 0; i < 2; ++i) {
        L.x[i] = b.take<float>(N * C);
        L.xh[i] = b.take<float>(N * C);
    }
    // CompressBlock
    L.c_win_vstart = b.take<int>(cap);
    L.c_win_cnt = b.take<int>(cap);
    L.c_pair_base = b.take<int>(cap);
    L.c_pair_vox = b.take<int>(cap);
    L.c_qp = b.take<float>(cap * C);
    L.c_ktok = b.take<float>(N * C);
    L.c_score = b.take<float>(N * (size_t)(f.has_cmp ? C / f.cmp.head_dim : 1));
    L.c_vp = b.take<float>(N * C);
    L.c_new = b.take<float>(cap * C);
    L.total = al256(b.off);
}

Frame *as_frame(void *h) { return reinterpret_cast<Frame *>(h); }

}  // namespace

extern "C" int mssvt_frame_destroy(void *frame);
extern "C" int mssvt_frame_create(void **frame_out) {
    if (!frame_out) return MSSVT_E_BADARG;
    Frame *f = new (std::nothrow) Frame();
    if (!f) return MSSVT_E_BADARG;
    hipError_t e = hipEventCreateWithFlags(&f->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&f->host_words), FR_HOST_WORDS * sizeof(int), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking);
    if (e != hipSuccess) {
        mssvt_frame_destroy(f);
        return (int)e;
    }
    *frame_out = f;
    return MSSVT_OK;
}

extern "C" int mssvt_frame_destroy(void *frame) {
    Frame *f = as_frame(frame);
    if (!f) return MSSVT_OK;
    if (f->ready) (void)hipEventDestroy(f->ready);
    if (f->fork) (void)hipEventDestroy(f->fork);
    if (f->join) (void)hipEventDestroy(f->join);
    if (f->side) (void)hipStreamDestroy(f->side);
    if (f->host_words) (void)hipHostFree(f->host_words);
    delete f;
    return MSSVT_OK;
}

extern "C" int mssvt_frame_set_overlap(void *frame, int on) {
    Frame *f = as_frame(frame);
    if (!f) return MSSVT_E_BADARG;
    f->overlap = on ? 1 : 0;
    return MSSVT_OK;
}

extern "C" int mssvt_frame_set_level(void *frame, int batch_size, int x_max, int y_max, int z_max, int hash_size,
                                     const float *host_voxel_size3, const float *host_range6, int C, int FF) {
    Frame *f = as_frame(frame);
    if (!f || !host_voxel_size3 || !host_range6 || batch_size <= 0 || x_max <= 0 || y_max <= 0 || z_max <= 0 || hash_size <= 0)
        return MSSVT_E_BADARG;
    if (z_max > 64 || mssvt_ffn_packed_bytes(C, FF) == 0) return MSSVT_E_TOOLARGE;  // sorted set-up; FFN shape instantiated
    f->B = batch_size; f->X = x_max; f->Y = y_max; f->Z = z_max; f->H = hash_size; f->C = C; f->FF = FF;
    for (int i = 0; i < 3; ++i) f->vs[i] = host_voxel_size3[i];
    for (int i = 0; i < 6; ++i) f->range[i] = host_range6[i];
    f->blocks.clear();
    f->has_cmp = false;
    f->level_set = true;
    return MSSVT_OK;
}

extern "C" int mssvt_frame_add_block(
    void *frame, const int *host_win1_size3, int max_num_odd, int max_num_even, int max_num_win1, int max_num_win2,
    int num_odd, int num_even, int num_win1, int num_win2, const int *vox_query_odd, const int *vox_query_even,
    const int *vox_query_win1, const int *vox_query_win2, const int *host_footprint4, const int *packed_offsets,
    int key_num_sample, int max_num_wins, int cbs_pattern, int use_interpolation, const float *norm1_w,
    const float *norm1_b, float norm1_eps, int num_groups, const int *host_c0, const int *host_cg, const int *host_heads,
    int head_dim, float scale, const float *const *host_Wq, const float *const *host_bq, const float *const *host_Wkv,
    const float *const *host_bkv, const float *const *host_Wo, const float *const *host_bo,
    const void *const *host_packed, const float *Wpos, const float *bpos, int attn_mode, const float *norm2_w,
    const float *norm2_b, float norm2_eps, const float *W1, const float *b1, const float *W2, const float *b2,
    const void *ffn_packed) {
    Frame *f = as_frame(frame);
    if (!f || !f->level_set || !host_win1_size3 || !vox_query_odd || !vox_query_even || !vox_query_win1 || !vox_query_win2 ||
        !host_footprint4 || !packed_offsets || !norm1_w || !norm1_b || !host_c0 || !host_cg || !host_heads || !host_Wq ||
        !host_bq || !host_Wkv || !host_bkv || !host_Wo || !host_bo || !Wpos || !bpos || !norm2_w || !norm2_b || !W1 || !b1 ||
        !W2 || !b2 || !ffn_packed)
        return MSSVT_E_BADARG;
    if (f->has_cmp) return MSSVT_E_TOOLARGE;  // one resolution level: Blocks, then the CompressBlock that ends it
    if (num_groups != 2 || cbs_pattern < 0 || cbs_pattern > 2 || attn_mode < 0 || attn_mode > 2 || key_num_sample > 64 ||
        host_footprint4[2] * host_footprint4[3] > 1024 || f->blocks.size() >= 16)
        return MSSVT_E_TOOLARGE;
    FrPlanCfg p;
    for (int i = 0; i < 3; ++i) p.ws[i] = host_win1_size3[i];
    p.n_o = max_num_odd; p.n_e = max_num_even; p.n1 = max_num_win1; p.n2 = max_num_win2;
    p.num_o = num_odd; p.num_e = num_even; p.num_1 = num_win1; p.num_2 = num_win2;
    p.K = key_num_sample; p.max_wins = max_num_wins;
    p.t_o = vox_query_odd; p.t_e = vox_query_even; p.t_1 = vox_query_win1; p.t_2 = vox_query_win2;
    p.packed_offsets = packed_offsets;
    for (int i = 0; i < 4; ++i) p.fp4[i] = host_footprint4[i];
    if (f->blocks.empty()) {
        f->plan = p;
    } else {
        // every Block of the level shares ONE window plan (the reference recomputes it per block, mssvt_backbone.py:139-199)
        const FrPlanCfg &q = f->plan;
        bool same = p.n_o == q.n_o && p.n_e == q.n_e && p.n1 == q.n1 && p.n2 == q.n2 && p.num_o == q.num_o && p.num_e == q.num_e &&
                    p.num_1 == q.num_1 && p.num_2 == q.num_2 && p.K == q.K && p.max_wins == q.max_wins;
        for (int i = 0; i < 3; ++i) same = same && p.ws[i] == q.ws[i];
        for (int i = 0; i < 4; ++i) same = same && p.fp4[i] == q.fp4[i];
        // (the offset tables are per-block copies of the same contents -- the caller compares them, fused.plan_key --
        // the first Block's are used)
        if (!same) return MSSVT_E_TOOLARGE;
    }
    FrBlock k;
    k.cbs_pattern = cbs_pattern; k.interp = use_interpolation ? 1 : 0;
    k.n1w = norm1_w; k.n1b = norm1_b; k.n1eps = norm1_eps;
    k.ng = num_groups; k.head_dim = head_dim; k.scale = scale;
    k.have_packed = host_packed != nullptr;
    for (int g = 0; g < 2; ++g) {
        k.c0[g] = host_c0[g]; k.cg[g] = host_cg[g]; k.heads[g] = host_heads[g];
        k.Wq[g] = host_Wq[g]; k.bq[g] = host_bq[g]; k.Wkv[g] = host_Wkv[g]; k.bkv[g] = host_bkv[g];
        k.Wo[g] = host_Wo[g]; k.bo[g] = host_bo[g];
        k.packed[g] = host_packed ? host_packed[g] : nullptr;
        if (!k.Wq[g] || !k.bq[g] || !k.Wkv[g] || !k.bkv[g] || !k.Wo[g] || !k.bo[g]) return MSSVT_E_BADARG;
    }
    if (!f->blocks.empty() && (k.cg[0] != f->blocks[0].cg[0] || k.cg[1] != f->blocks[0].cg[1] ||
                               k.heads[0] != f->blocks[0].heads[0] || k.heads[1] != f->blocks[0].heads[1]))
        return MSSVT_E_TOOLARGE;  // one hand-off scratch for all Blocks
    k.Wp = Wpos; k.bp = bpos; k.attn_mode = attn_mode;
    k.n2w = norm2_w; k.n2b = norm2_b; k.n2eps = norm2_eps;
    k.W1 = W1; k.b1 = b1; k.W2 = W2; k.b2 = b2; k.ffn_packed = ffn_packed;
    // at most 3 query patterns / 4 interpolation tables per plan (mssvt_window_plan_two)
    int n_tabs = 0, tp[8], ti[8];
    std::vector<FrBlock> all = f->blocks;
    all.push_back(k);
    for (const FrBlock &q : all) {
        bool seen = false;
        for (int t = 0; t < n_tabs; ++t) seen = seen || (tp[t] == q.cbs_pattern && ti[t] == q.interp);
        if (!seen) {
            if (n_tabs == 4) return MSSVT_E_TOOLARGE;
            tp[n_tabs] = q.cbs_pattern;
            ti[n_tabs++] = q.interp;
        }
    }
    f->blocks.push_back(k);
    return MSSVT_OK;
}

extern "C" int mssvt_frame_add_compress(
    void *frame, const int *host_win_size3, int max_num_win1, int num_win1, const int *vox_query_win1, int max_num_wins,
    const float *norm1_w, const float *norm1_b, float norm1_eps, const float *Wpos1, const float *bpos1, const float *Wpos2,
    const float *bpos2, const float *Wq, const float *bq, const float *Wkv, const float *bkv, const float *Wo, const float *bo,
    int head_dim, float scale, int split_f16, const float *norm2_w, const float *norm2_b, float norm2_eps, const float *W1,
    const float *b1, const float *W2, const float *b2, const void *ffn_packed, const void *compress_ws_packed) {
    Frame *f = as_frame(frame);
    if (!f || !f->level_set || !host_win_size3 || !vox_query_win1 || !norm1_w || !norm1_b || !Wpos1 || !bpos1 || !Wpos2 ||
        !bpos2 || !Wq || !bq || !Wkv || !bkv || !Wo || !bo || !norm2_w || !norm2_b || !W1 || !b1 || !W2 || !b2 || !ffn_packed)
        return MSSVT_E_BADARG;
    // pillar windows of a sorted level (one lane per window in the plan kernel), one head group
    if (f->has_cmp || f->blocks.empty() || host_win_size3[0] != 1 || host_win_size3[1] != 1 || num_win1 < 1 || num_win1 > 64 ||
        head_dim <= 0 || f->C % head_dim != 0)
        return MSSVT_E_TOOLARGE;
    FrCompress &c = f->cmp;
    for (int i = 0; i < 3; ++i) c.ws[i] = host_win_size3[i];
    c.ns = max_num_win1; c.num_1 = num_win1; c.t_1 = vox_query_win1; c.max_wins = max_num_wins;
    c.n1w = norm1_w; c.n1b = norm1_b; c.n1eps = norm1_eps;
    c.Wp1 = Wpos1; c.bp1 = bpos1; c.Wp2 = Wpos2; c.bp2 = bpos2;
    c.Wq = Wq; c.bq = bq; c.Wkv = Wkv; c.bkv = bkv; c.Wo = Wo; c.bo = bo;
    c.head_dim = head_dim; c.scale = scale; c.split_f16 = split_f16;
    c.n2w = norm2_w; c.n2b = norm2_b; c.n2eps = norm2_eps;
    c.W1 = W1; c.b1 = b1; c.W2 = W2; c.b2 = b2; c.ffn_packed = ffn_packed;
    // one launch instead of three when the caller prepared the fragments (it has checked: every cell of the slab listed, fp16
    // range) and no list can be truncated -- a window is then one run of rows of the sorted level
    c.ws_packed = compress_ws_packed && f->C == 128 && head_dim == 16 && host_win_size3[2] <= max_num_win1 && max_num_win1 <= 32
                      ? compress_ws_packed : nullptr;
    f->has_cmp = true;
    return MSSVT_OK;
}

extern "C" long long mssvt_frame_workspace_bytes(void *frame, int num_voxels) {
    Frame *f = as_frame(frame);
    if (!f || !f->level_set || !f->has_cmp || num_voxels < 0) return 0;
    Layout L;
    make_layout(*f, num_voxels, nullptr, L);
    return (long long)L.total;
}

#define FR_TRY(call_)                  \
    {                                  \
        const int st_ = (call_);       \
        if (st_ != MSSVT_OK) return st_; \
    }

extern "C" int mssvt_frame_forward(void *frame, int num_voxels, const float *features, const int *indices, void *workspace,
                                   long long workspace_bytes, float *out_features, int *out_indices, int *out_table,
                                   int *out_counts, void *stream_) {
    Frame *f = as_frame(frame);
    if (!f || !f->level_set || !f->has_cmp || f->blocks.empty() || num_voxels <= 0 || !features || !indices || !workspace ||
        !out_features || !out_indices || !out_table || !out_counts || ((uintptr_t)workspace & 255) || ((uintptr_t)out_table & 15))
        return MSSVT_E_BADARG;
    Layout L;
    make_layout(*f, num_voxels, reinterpret_cast<char *>(workspace), L);
    if ((long long)L.total > workspace_bytes) return MSSVT_E_BADARG;
    hipStream_t stream = (hipStream_t)stream_;
    const int n = num_voxels, B = f->B, X = f->X, Y = f->Y, Z = f->Z, H = f->H, C = f->C, FF = f->FF, cap = n;
    const FrPlanCfg &p = f->plan;
    const FrCompress &c = f->cmp;

    // ---- one fill: -1 arena, the output level's hash table, zero region, the attention buffers' zero row
    bool ln_in_fill = false;
    {
        const long long n_tab = (long long)B * H * 2, n_zr = (C + 3) / 4 * 4;
        const long long quads = ((long long)L.neg_ints + n_tab + (long long)L.zero_ints + n_zr) >> 2;
        int grid = (int)((quads + 255) / 256 > 4096 ? 4096 : (quads + 255) / 256);
        if (grid < 1) grid = 1;
        // ... and the first norm1 in the same launch (it depends on nothing the fill writes) unless it goes to the side stream
        const FrBlock &k0 = f->blocks[0];
        const int lpr = C / 4, ln_blocks = (n + 4 * (MSSVT_WAVE / (lpr > 0 ? lpr : 1)) - 1) / (4 * (MSSVT_WAVE / (lpr > 0 ? lpr : 1)));
        ln_in_fill = !f->overlap && (lpr == 8 || lpr == 16 || lpr == 32);
#define FR_FILL_LN(LPR_)                                                                                                  \
    k_frame_fill_ln<LPR_><<<grid + ln_blocks, 256, 0, stream>>>(L.neg0, (long long)L.neg_ints, out_table, n_tab, L.zero0, \
                                                                (long long)L.zero_ints, reinterpret_cast<int *>(L.attn_zero), \
                                                                n_zr, grid, features, n, k0.n1w, k0.n1b, k0.n1eps, L.xhat0)
        if (ln_in_fill && lpr == 32) FR_FILL_LN(32);
        else if (ln_in_fill && lpr == 16) FR_FILL_LN(16);
        else if (ln_in_fill && lpr == 8) FR_FILL_LN(8);
        else
            k_frame_fill<<<grid, 256, 0, stream>>>(L.neg0, (long long)L.neg_ints, out_table, n_tab, L.zero0, (long long)L.zero_ints,
                                                   reinterpret_cast<int *>(L.attn_zero), n_zr);
#undef FR_FILL_LN
        FR_TRY(mssvt_launch_status());
    }
    // ---- level set-up (sorted list; the device verifies the order): partition 0 = the Blocks' windows, 1 = the pillars
    const int grid3[6] = {X / p.ws[0], Y / p.ws[1], Z / p.ws[2], X / c.ws[0], Y / c.ws[1], Z / c.ws[2]};
    const int wsize3[6] = {p.ws[0], p.ws[1], p.ws[2], c.ws[0], c.ws[1], c.ws[2]};
    const int maxw[2] = {p.max_wins, c.max_wins};
    int *wins[2] = {L.win_blk, out_indices}, *tables[2] = {nullptr, out_table}, *vcounts[2] = {L.vcount_blk, out_counts};
    int *hdrs[2] = {L.hdr[0], L.hdr[1]};
    // (the CompressBlock's pillar lists fall out of the same launches: every pillar window is a slab of one column word)
    FR_TRY(mssvt_level_setup_sorted_pillars(n, B, X, Y, Z, H, indices, L.zero0, -(long long)L.zero_ints * 4, L.cnt, L.start, L.occ,
                                            L.vbase, L.status, 2, grid3, wsize3, maxw, wins, tables, vcounts, hdrs, L.scratch, 1,
                                            c.ns, c.num_1, c.t_1, L.c_k_ind, L.c_win_vstart, L.c_win_cnt, nullptr, L.pair_win,
                                            nullptr, stream));
    // the words the host needs (level status | per partition: status, window count) are final here: copy them out now
    f->words = 64 * 3;
    {
        // (round 6 measured the copy on the side stream, behind an event, to take its ~4 us off the critical path: 1 526 against
        // 1 542 frames/s one frame at a time -- the cross-queue event costs more than the copy; it stays in stream order)
        hipError_t e = hipMemcpyAsync(f->host_words, L.status, (size_t)f->words * sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipEventRecord(f->ready, stream);
        if (e != hipSuccess) return (int)e;
    }
    // ---- first norm1: on the side stream under the plan kernel when asked to.  Between the fork and the join every error
    // return goes through `unfork` (the side stream may still be writing L.xhat0 of the caller's workspace)
    hipStream_t s2 = stream;
    int forked = 0;  // 1: side stream has work, join not recorded; 2: join recorded, not yet waited for
    auto unfork = [&](int rc) {
        if (forked == 2 && hipStreamWaitEvent(stream, f->join, 0) == hipSuccess) forked = 0;
        if (forked) (void)hipStreamSynchronize(f->side);
        forked = 0;
        return rc;
    };
#define FR_TRY_FORKED(call_)                          \
    {                                                 \
        const int st_ = (call_);                      \
        if (st_ != MSSVT_OK) return unfork(st_);      \
    }
    if (f->overlap) {
        hipError_t e = hipEventRecord(f->fork, stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(f->side, f->fork, 0);
        if (e != hipSuccess) return (int)e;
        s2 = f->side;
        forked = 1;
    }
    if (!ln_in_fill)
        FR_TRY_FORKED(mssvt_layer_norm(features, n, C, f->blocks[0].n1w, f->blocks[0].n1b, f->blocks[0].n1eps, L.xhat0, s2));
    if (f->overlap) {
        const hipError_t e = hipEventRecord(f->join, s2);
        if (e != hipSuccess) return unfork((int)e);
        forked = 2;
    }
    // ---- window plan of the Blocks, with the interpolation tables of every (pattern, interpolation) variant
    const float mn3[3] = {f->range[0], f->range[1], f->range[2]};
    const float wsm[3] = {f->vs[0] * p.ws[0], f->vs[1] * p.ws[1], f->vs[2] * p.ws[2]};
    {
        int tab_list[4], tab_zero[4];
        int *tab_row[4];
        float *tab_w[4];
        for (int t = 0; t < L.n_tabs; ++t) {
            tab_list[t] = q_list(L.tab_pat[t]);
            tab_zero[t] = L.attn_zero_row[pattern_slot(L, L.tab_pat[t])];
            tab_row[t] = L.tab_rows + (size_t)t * n * 4;
            tab_w[t] = L.tab_w + (size_t)t * n * 4;
        }
        FR_TRY_FORKED(mssvt_window_plan_two(
            X, Y, Z, p.ws[0], p.ws[1], p.ws[2], p.n_o, p.n_e, p.n1, p.n2, H, B, p.num_o, p.num_e, p.num_1, p.num_2, p.t_o, p.t_e,
            p.t_1, p.t_2, p.K, L.win_blk, L.hdr[0] + 1, cap, nullptr, L.cnt, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
            nullptr, L.win_vstart, nullptr, nullptr, nullptr, indices, f->vs, mn3, wsm,  // (lists / key indices / owners: nobody reads them here)
            L.qmeta[0], L.qmeta[1], L.qmeta[2], L.kmeta[0], L.kmeta[1], L.wcentre, L.nq_valid, L.occ, p.fp4, p.packed_offsets,
            L.vbase, L.status, L.vcount_blk, L.n_tabs, tab_list, L.tab_interp, tab_zero, tab_row, tab_w, stream));
    }
    // ---- work orders + compact query rows of every query pattern in one launch group
    {
        const int *nqv[3];
        int nq[3];
        const float *qm[3];
        for (int i = 0; i < L.n_pat; ++i) {
            const int l = q_list(L.pats[i]);
            nqv[i] = L.nq_valid + (size_t)l * cap;
            nq[i] = q_slots(p, L.pats[i]);
            qm[i] = L.qmeta[l];
        }
        // (counting the order histograms inside the plan kernel with atomics was measured: the plan kernel +16.7 us for the
        // 4.9 us launch it saves)
        FR_TRY_FORKED(mssvt_plan_order_multi(L.n_pat, L.hdr[0] + 1, nqv, nq, qm, cap, (int)L.row_cap, L.perm, L.n_act, L.q_off, L.row_meta,
                                      L.row_src, L.n_rows, stream));
    }
    if (f->overlap) {
        const hipError_t e = hipStreamWaitEvent(stream, f->join, 0);
        if (e != hipSuccess) return unfork((int)e);
        forked = 0;
    }
#undef FR_TRY_FORKED
    // ---- the Blocks
    const float *x = features, *xhat = L.xhat0;
    const int nb = (int)f->blocks.size();
    for (int i = 0; i < nb; ++i) {
        const FrBlock &k = f->blocks[i];
        const int s = pattern_slot(L, k.cbs_pattern), l = q_list(k.cbs_pattern), nq = q_slots(p, k.cbs_pattern);
        const float *kmeta[2] = {L.kmeta[0], L.kmeta[1]};
        const int *nqv = L.nq_valid + (size_t)l * cap;
        if (k.attn_mode == 2) {
            FR_TRY(mssvt_block_attention_bf16(C, k.ng, k.c0, k.cg, k.heads, k.head_dim, k.scale, nq, p.K, xhat, L.n_act[s], L.perm[s],
                                              L.q_off[s], nqv, L.n_rows[s], (int)L.row_cap, L.row_meta[s], L.row_src[s], kmeta,
                                              L.wcentre, k.Wq, k.bq, k.Wkv, k.bkv, k.Wo, k.bo, k.Wp, k.bp, L.attn[s], stream));
        } else if (k.attn_mode == 1) {
            FR_TRY(mssvt_block_attention_kv16(C, k.ng, k.c0, k.cg, k.heads, k.head_dim, k.scale, nq, p.K, xhat, L.n_act[s], L.perm[s],
                                              L.q_off[s], nqv, L.n_rows[s], (int)L.row_cap, L.row_meta[s], L.row_src[s], kmeta,
                                              L.wcentre, k.Wq, k.bq, k.Wkv, k.bkv, k.Wo, k.bo, k.Wp, k.bp, L.qbuf, L.attn[s],
                                              k.have_packed ? k.packed : nullptr, stream));
        } else {
            FR_TRY(mssvt_block_attention(C, k.ng, k.c0, k.cg, k.heads, k.head_dim, k.scale, nq, p.K, xhat, L.n_act[s], L.perm[s],
                                         L.q_off[s], nqv, L.n_rows[s], (int)L.row_cap, L.row_meta[s], L.row_src[s], kmeta, L.wcentre,
                                         k.Wq, k.bq, k.Wkv, k.bkv, k.Wo, k.bo, k.Wp, k.bp, L.qbuf, L.attn[s], stream));
        }
        int t = 0;
        while (t < L.n_tabs && !(L.tab_pat[t] == k.cbs_pattern && L.tab_interp[t] == k.interp)) ++t;
        // the FFN tail emits the NEXT block's norm1 (the CompressBlock's after the last Block)
        const float *nw = i + 1 < nb ? f->blocks[i + 1].n1w : c.n1w, *nbias = i + 1 < nb ? f->blocks[i + 1].n1b : c.n1b;
        const float neps = i + 1 < nb ? f->blocks[i + 1].n1eps : c.n1eps;
        // (the last Block's y has no reader: the CompressBlock takes the LayerNorm output only -- not stored)
        float *y = i + 1 < nb ? L.x[i & 1] : nullptr, *yn = L.xh[i & 1];
        FR_TRY(mssvt_ffn_fused_interp(n, C, FF, x, L.tab_rows + (size_t)t * n * 4, L.tab_w + (size_t)t * n * 4, L.attn[s], k.n2w,
                                      k.n2b, k.n2eps, k.W1, k.b1, k.W2, k.b2, y, nw, nbias, neps, yn,
                                      const_cast<float *>(reinterpret_cast<const float *>(k.ffn_packed)), nullptr, 4, stream));
        x = y;
        xhat = yn;
    }
    // ---- CompressBlock: attention (one launch on a sorted pillar level, else three), FFN tail over the live windows
    const float cwsm[3] = {f->vs[0] * c.ws[0], f->vs[1] * c.ws[1], f->vs[2] * c.ws[2]};
    if (c.ws_packed) {
        FR_TRY(mssvt_compress_ws(C, c.head_dim, c.scale, c.ws[2], c.ns, n, L.hdr[1] + 1, cap, indices, L.c_win_cnt, L.pair_win,
                                 f->vs, mn3, cwsm, xhat, c.Wp1, c.bp1, c.bp2, c.bq, c.bkv, c.bo, c.ws_packed,
                                 L.c_new, stream));
    } else {
        FR_TRY(mssvt_compress_fused(C, c.head_dim, c.scale, c.ns, n, L.hdr[1] + 1, cap, out_indices, indices, L.c_k_ind, L.c_win_vstart,
                                    L.c_win_cnt, L.pair_win, f->vs, mn3, cwsm, xhat, c.Wp1, c.bp1, c.Wp2, c.bp2, c.Wq, c.bq, c.Wkv,
                                    c.bkv, c.Wo, c.bo, L.c_qp, L.c_ktok, L.c_score, L.c_vp, L.c_new, c.split_f16, stream));
    }
    FR_TRY(mssvt_ffn_fused(cap, C, FF, L.c_new, nullptr, nullptr, c.n2w, c.n2b, c.n2eps, c.W1, c.b1, c.W2, c.b2, out_features, nullptr,
                           nullptr, 0.f, nullptr, const_cast<float *>(reinterpret_cast<const float *>(c.ffn_packed)), L.hdr[1] + 1, 4,
                           stream));
    return MSSVT_OK;
}

extern "C" int mssvt_frame_wait_words(void *frame, int *host_out, int num_words) {
    Frame *f = as_frame(frame);
    if (!f || !host_out || num_words <= 0 || num_words > f->words) return MSSVT_E_BADARG;
    const hipError_t e = hipEventSynchronize(f->ready);
    if (e != hipSuccess) return (int)e;
    for (int i = 0; i < num_words; ++i) host_out[i] = f->host_words[i];
    return MSSVT_OK;
}
