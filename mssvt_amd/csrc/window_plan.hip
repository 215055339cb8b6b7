// window_plan.hip -- fused, device-resident "window plan" for the module fast path.
//
// One launch does, for every non-empty window, what the reference does with
// K3 (gather_two_window_voxels, ref: mssvt/src/ms_sparse_attention_gpu.cu:193-350),
// two K7 launches (farthest point sampling of 32 keys from the win1 and win2 lists,
// ref: pointnet2/pointnet2_batch/src/sampling_gpu.cu:100-216), two K8 launches plus
// the mask logic of mssvt_backbone.py:247-258 -- without ever writing the padded
// (nw,343) index / (nw,343,3) offset arrays to HBM (~130 MB per call at 160k
// points in the reference): the hit lists live in LDS, FPS runs on them in place,
// and only what the attention stage consumes is stored:
//     ind_odd / ind_even / ind_win1   (-1 padded, identical to K3's outputs)
//     k_ind1 / k_ind2                 the 32 sampled key voxels per scale
//     k_mask1 / k_mask2               1 = masked key slot
//     win_vstart                      first feature row of the window's sample
//     owner_*                         per voxel, the flat list slot that updates it
// One WAVEFRONT per window; lanes probe 64 offsets at a time (ballot + prefix
// popcount keeps the reference's append order), then replay the reference's FPS
// block tree so that ties resolve identically (see fps_nn.hip).
//
// The number of windows is read from DEVICE memory (no host round trip); the grid
// is sized for the capacity and surplus waves exit.
#include <stdlib.h>
#include "common.hip.h"
#define PLAN_MAX_WPB 4
#ifndef PLAN_WAVES_PER_SIMD
#define PLAN_WAVES_PER_SIMD 5  // register budget of the common instantiation (96 VGPRs; 8, 6 and 4 waves measured the same)
#endif

struct PlanArgs {
    int x_max, y_max, z_max, x_ws, y_ws, z_ws;
    int max_odd, max_even, max_win1, max_win2;
    int hash_size, batch_size;
    int n_odd, n_even, n_win1, n_win2;
    const int *q_odd, *q_even, *q_win1, *q_win2;
    int key_num_sample, bs1, bs2;  // FPS picks, reference block sizes for n = max_win1 / max_win2
    const int *win_indices;
    const int *num_wins;  // device scalar
    const slot_t *table;
    const int *v_bs_cnt;
    int *ind_odd, *ind_even, *ind_win1, *k_ind1, *k_ind2;
    unsigned char *k_mask1, *k_mask2;
    int *win_vstart;
    int *owner_win1, *owner_odd, *owner_even;
    // resolved metadata for the attention kernel (optional, all or none): per slot a float4
    // (rel.x, rel.y, rel.z, bits(global feature row or -1)); rel = voxel centre - window centre in
    // metres, computed exactly as ref with_coords does (mssvt_backbone.py:132-137, :269-276)
    float4 *qmeta_odd, *qmeta_even, *qmeta_win1, *kmeta1, *kmeta2, *wcentre;
    int *nq_valid;  // (3, cap): valid odd / even / win1 entries per window (work estimate)
    int win_capacity;
    const int *indices;
    float vsx, vsy, vsz, minx, miny, minz, wsx, wsy, wsz;
    int lds_words_per_wave;
    int hit_cap;   // entries of the per-wave hit sequence (>= number of table offsets, even)
    int meta_cap;  // hits whose resolved metadata is kept in LDS (covers the odd / even / win1 lists)
    // optional occupancy columns (mssvt_occupancy_columns): one 64-bit word per (b, x, y), bit z set
    // when the cell holds a voxel; fx0/fy0/fnx/fny = bounding box of the tables' (x, y) offsets
    const unsigned long long *occ;
    int fx0, fy0, fnx, fny, fny_magic;  // (c * fny_magic) >> 20 == c / fny for every column c of the footprint
    const int *q_packed;  // with occ: the four tables concatenated, one pack_off() word per offset
    // optional, with occ: voxels of the sample in the columns before (b, x, y) of a voxel list that is sorted by
    // (b, x, y, z) -- the index of an occupied cell is then col_vbase + popcount(column word below z) and the hash
    // is not probed at all; level_status = the device word in which mssvt_level_setup_sorted reports ST_UNSORTED
    const int *col_vbase, *level_status;
    const int *win_counts;  // (unused: a centre-out work order was measured and bought nothing)
    // interpolation tables (ref K9 + K10 + weights, mssvt_backbone.py:298-311; csrc/block_attn.hip k_block_scatter is the
    // stand-alone form): per voxel the three attention rows + weights its update comes from, for up to PLAN_MAX_TABS
    // (query list, interpolation) variants -- built here, where the window's lists sit in LDS, when the lists of
    // different windows cannot overlap (every voxel then has one owner: this window)
    int n_tabs, tab_q;  // tab_q: candidate slots per wave (largest query list + 3)
    struct PlanTab {
        int list, maxn, interp, zero_row;  // list: 0 odd, 1 even, 2 win1 (the queries); zero_row: attention row of zeros
        int4 *tab_row;
        float4 *tab_w;
    } tabs[4];
};

__device__ __forceinline__ float plan_centre(int idx, float cell, float lo) {
    return __fadd_rn(__fmul_rn(__fadd_rn((float)idx, 0.5f), cell), lo);
}

// one word per table offset: (x + 64) | (y + 64) << 7 | (z + 64) << 14 | column << 21, column = (x - fx0) * fny + (y - fy0)
// = the offset's (x, y) column inside the tables' footprint (0 without occupancy columns)
#define PACK0 (64 | (64 << 7) | (64 << 14))
#define PLAN_PRE 6  // 64-offset steps of K3 whose packed offsets are preloaded into registers (7 x 7 x 7: 343 offsets)
__device__ __forceinline__ int pack_off(int ox, int oy, int oz) {
    return (ox + 64) | ((oy + 64) << 7) | ((oz + 64) << 14);
}

// ---- farthest point sampling on the hit sequence ---------------------------------------------------------
// (ref pointnet2/pointnet2_batch/src/sampling_gpu.cu:93-216 replayed; see fps_nn.hip for the operator form)
//
// Offsets are integers in [-60, 60]: every squared distance is an integer < 2^16, exact in fp32, so the sampler
// runs on integers with the same outcome: |a - b|^2 = |a|^2 - 2 a.b + |b|^2 is ONE v_dot4_i32_i8 (a and -2 b packed
// as signed bytes) plus one add.  One 32-bit key
//     (distance << 12) | ((1023 - bit-reversed thread id) << 2) | (first slot of the thread ? 2 : 0) | 1
// orders candidates exactly like the reference's block: larger distance first; among equal distances the
// shared-memory tree keeps the lower thread at every stride bs/2 ... 1, i.e. the smallest BIT-REVERSED thread id;
// inside a thread the later slot wins only on a STRICTLY larger distance (bit 1).  0 = no candidate.  One wave
// max per round; the winner's thread id and slot are decoded from the key itself (scalar), no ballot / readlane.
#define FPS_BIG 0xFFFFFu  // "1e10": replaced by a real distance in the first round
template <int ROWS = 4>  // candidates only in lanes [0, 16 ROWS)
__device__ __forceinline__ unsigned int wave_max_u32_uniform(unsigned int v) {
    unsigned int t;
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); v = t > v ? t : v;   // quad_perm [1,0,3,2]
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); v = t > v ? t : v;   // quad_perm [2,3,0,1]
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); v = t > v ? t : v;  // row_half_mirror
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); v = t > v ? t : v;  // row_mirror
    if (ROWS == 1) return (unsigned int)__builtin_amdgcn_readlane((int)v, 15);
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); v = t > v ? t : v;  // row_bcast:15
    if (ROWS == 2) return (unsigned int)__builtin_amdgcn_readlane((int)v, 31);
    t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); v = t > v ? t : v;  // row_bcast:31
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
// signed-byte packs of an offset (x, y, z) and of -2 (x, y, z) (|offset| <= 60)
__device__ __forceinline__ int fps_pack_s8(int x, int y, int z) { return (x & 255) | ((y & 255) << 8) | ((z & 255) << 16); }
__device__ __forceinline__ void fps_unpack(int pk, int &x, int &y, int &z) {
    x = (pk & 127) - 64; y = ((pk >> 7) & 127) - 64; z = ((pk >> 14) & 127) - 64;
}
// thread of the reference block with the smallest bit-reversed id in [lo, bs): the one with the most trailing zeros
__device__ __forceinline__ int fps_min_rev_thread(int lo, int L) {
    for (int t = L - 1; t >= 0; --t) {
        const int c = (lo + (1 << t) - 1) & ~((1 << t) - 1);
        if (c < (1 << L)) return c;
    }
    return lo;
}
__device__ __forceinline__ int fps_decode_pick(unsigned int mkey, int L, int bs) {
    const unsigned int rev = 1023u - ((mkey >> 2) & 1023u);
    const int vt = L ? (int)(__brev(rev) >> (32 - L)) : 0;
    return vt + ((mkey & 2u) ? 0 : bs);
}

// Fast path, the case that matters: the list holds nv <= 64 valid entries (front-packed, as K3 produces
// them) followed by padding, and nv <= bs.  All padding slots carry the same offset (0,0,0), hence the same running
// min-distance `tpad`, so nothing needs to be scanned per slot.  The reference block is reproduced analytically:
//   * reference thread vt visits slots vt, vt+bs (n <= 2*bs - 1): for vt < nv that is one valid slot
//     then (if it exists) one padding slot, taken only if tpad is STRICTLY larger; for vt >= nv it
//     is padding only -> (tpad, vt);
//   * among the pure-padding threads [nv, bs) the tree favours the smallest bit-reversed id (fps_min_rev_thread).
// One lane per valid entry (pk = its packed offset, PACK0 on the other lanes), min-distances in registers.
template <bool REGPICKS, int ROWS>
__device__ __forceinline__ void fps_on_list_fast(int pk, int n, int nv, int m, int bs, int *fps_out, int lane) {
    const int L = 31 - __clz(bs);
    const bool mine = lane < nv;
    int xk, yk, zk;
    fps_unpack(pk, xk, yk, zk);
    const int a4 = fps_pack_s8(xk, yk, zk), m2 = fps_pack_s8(-2 * xk, -2 * yk, -2 * zk);
    const int na = xk * xk + yk * yk + zk * zk;
    const unsigned int myrev = __brev((unsigned int)lane) >> (32 - L);
    const unsigned int tb = mine ? (((1023u - myrev) << 2) | 3u) : 0u;  // first slot of thread `lane`
    const bool has_second = mine && lane + bs < n;                      // this thread's second slot (padding)
    const unsigned int tb2 = has_second ? (tb & ~2u) : 0u, hs_mask = has_second ? 0xFFFFFFFFu : 0u;
    const bool padgroup = nv < bs;  // pure-padding threads vt in [nv, bs)
    const int vp = padgroup ? fps_min_rev_thread(nv, L) : 0;
    const unsigned int vp_tb = padgroup ? (((1023u - (__brev((unsigned int)vp) >> (32 - L))) << 2) | 3u) : 0u;
    const unsigned int pgmask = padgroup ? 0xFFFFFFFFu : 0u;
    // no padding slot at all (nv == n): tpad is never looked at -> 0 keeps the exit test to one compare
    unsigned int tk = mine ? FPS_BIG : 0u, tpad = nv < n ? FPS_BIG : 0u;
    int old = 0, picks = 0;
    if (!REGPICKS && lane == 0) fps_out[0] = 0;
    int j = 1;
    for (; j < m; ++j) {
        // offset of the last pick: a valid entry's, or (0,0,0) of a padding slot (`old` is wave-uniform)
        const bool ov = old < nv;
        const int ol = ov ? old : 0;
        int s_m2 = __builtin_amdgcn_readlane(m2, ol), s_nb = __builtin_amdgcn_readlane(na, ol);
        s_m2 = ov ? s_m2 : 0;
        s_nb = ov ? s_nb : 0;
        const unsigned int d = (unsigned int)(__builtin_amdgcn_sdot4(a4, s_m2, na, false) + s_nb);
        tk = d < tk ? d : tk;
        tpad = (unsigned int)s_nb < tpad ? (unsigned int)s_nb : tpad;
        const unsigned int tp12 = tpad << 12;
        const unsigned int k1 = (tk << 12) | tb;
        const unsigned int k2 = (tp12 & hs_mask) | tb2;
        const unsigned int key = k1 > k2 ? k1 : k2;
        const unsigned int mkey = wave_max_u32_uniform<ROWS>(key);
        const int dec = fps_decode_pick(mkey, L, bs);
        old = ((tp12 | vp_tb) & pgmask) > mkey ? vp : dec;  // a pure-padding thread outranks every entry
        if (REGPICKS)
            picks = lane == j ? old : picks;
        else if (lane == 0)
            fps_out[j] = old;
        // every remaining min-distance is 0: all further rounds tie completely and return slot 0
        if ((mkey | tp12) < 4096u) break;
    }
    if (REGPICKS) {
        if (lane < m) fps_out[lane] = lane <= j ? picks : 0;
    } else {
        for (int jj = j + 1 + lane; jj < m; jj += MSSVT_WAVE) fps_out[jj] = 0;
    }
}

// Register form of the same sampler for lists of any fill (bs <= 64 TPL, n < 2 bs): the reference
// block's thread vt = lane + 64 k owns slots vt and vt + bs, both kept in registers with their
// running min-distances -- padding slots (>= nv) are ordinary slots at offset (0,0,0).  Per round: update
// 2 TPL distances, reduce the lane's TPL threads and then the wave, fetch the picked slot's offset from the LDS
// list.  No LDS tree, no per-slot LDS traffic (10 % of the windows of a 160k-point scene hold > 64 entries and
// used to set the run time of the whole plan kernel).
template <int TPL>
__device__ __forceinline__ void fps_on_list_regs(const int *packed, int nv, int n, int m, int bs, int *fps_out, int lane) {
    const int L = 31 - __clz(bs);
    int a4[TPL], b4[TPL], na[TPL], nb[TPL];
    unsigned int da[TPL], db[TPL], tba[TPL], tbb[TPL];
#pragma unroll
    for (int k = 0; k < TPL; ++k) {
        const int vt = lane + MSSVT_WAVE * k;
        const bool ea = vt < bs && vt < n, eb = vt < bs && vt + bs < n;
        const int pa = (ea && vt < nv) ? packed[vt] : PACK0, pb = (eb && vt + bs < nv) ? packed[vt + bs] : PACK0;
        int x, y, z;
        fps_unpack(pa, x, y, z);
        a4[k] = fps_pack_s8(x, y, z); na[k] = x * x + y * y + z * z;
        fps_unpack(pb, x, y, z);
        b4[k] = fps_pack_s8(x, y, z); nb[k] = x * x + y * y + z * z;
        const unsigned int rev = L ? __brev((unsigned int)vt) >> (32 - L) : 0u;
        tba[k] = ea ? (((1023u - rev) << 2) | 3u) : 0u;
        tbb[k] = eb ? (((1023u - rev) << 2) | 1u) : 0u;
        da[k] = ea ? FPS_BIG : 0u;
        db[k] = eb ? FPS_BIG : 0u;
    }
    int old = 0;
    if (lane == 0) fps_out[0] = 0;
    for (int j = 1; j < m; ++j) {
        const int po = __builtin_amdgcn_readfirstlane(old < nv ? packed[old] : PACK0);
        int x1, y1, z1;
        fps_unpack(po, x1, y1, z1);
        const int s_m2 = fps_pack_s8(-2 * x1, -2 * y1, -2 * z1), s_nb = x1 * x1 + y1 * y1 + z1 * z1;
        unsigned int bestkey = 0u;
#pragma unroll
        for (int k = 0; k < TPL; ++k) {
            unsigned int d = (unsigned int)(__builtin_amdgcn_sdot4(a4[k], s_m2, na[k], false) + s_nb);
            da[k] = d < da[k] ? d : da[k];
            d = (unsigned int)(__builtin_amdgcn_sdot4(b4[k], s_m2, nb[k], false) + s_nb);
            db[k] = d < db[k] ? d : db[k];
            const unsigned int ka = (da[k] << 12) | tba[k], kb = tbb[k] ? ((db[k] << 12) | tbb[k]) : 0u;
            const unsigned int kk = ka > kb ? ka : kb;
            bestkey = kk > bestkey ? kk : bestkey;
        }
        const unsigned int mkey = wave_max_u32_uniform(bestkey);
        old = fps_decode_pick(mkey, L, bs);
        if (lane == 0) fps_out[j] = old;
        if ((mkey >> 12) == 0u) {
            // every min-distance is 0 from here on: all further rounds tie completely -> thread 0, slot 0
            for (int jj = j + 1 + lane; jj < m; jj += MSSVT_WAVE) fps_out[jj] = 0;
            break;
        }
    }
}

#ifdef MSSVT_STAMPS  // developer instrumentation
__device__ unsigned long long g_plan_stamps[32768 * 16];
__device__ unsigned long long g_plan_span[32768 * 2];
extern "C" int mssvt_debug_read_plan_span(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_plan_span), sizeof(g_plan_span));
}
extern "C" int mssvt_debug_read_plan_stamps(unsigned long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_plan_stamps), sizeof(g_plan_stamps));
}
#define PSTAMP()                                                                                   \
    {                                                                                              \
        if (lane == 0 && si < 16 && w < 32768) g_plan_stamps[w * 16 + si] = __builtin_readcyclecounter(); \
        ++si;                                                                                      \
    }
#else
#define PSTAMP()
#endif

// lanes below k as a 64-bit mask (wave-uniform k: scalar ALU)
__device__ __forceinline__ unsigned long long lanes_below(int k) {
    return k <= 0 ? 0ull : k >= 64 ? ~0ull : (1ull << k) - 1ull;
}

// One list of a window (K3's vox_ind_* row, ref ms_sparse_attention_gpu.cu:238-262): entries [first, first + nv) of
// the hit sequence, -1 padded to maxn; its resolved metadata (copied from the per-hit rows in LDS; optional); the owner
// array of its voxels.
__device__ __forceinline__ void plan_emit_list(int w, int lane, int maxn, int nv, int first, const int *hsv, const float4 *hmeta,
                                               int *ind, float4 *qmeta, int *owner, int vstart) {
    if (!ind && !qmeta && !owner) return;  // (nobody wants this list's rows: the whole-frame call skips what only the operator-level tests read)
    for (int k = lane; k < maxn; k += MSSVT_WAVE) {
        const bool valid = k < nv;
        const int e = valid ? first + k : 0;
        const int sv = valid ? hsv[e] : MSSVT_EMPTY;
        if (ind) ind[(size_t)w * maxn + k] = sv;
        if (qmeta) {
            const float4 m = hmeta[e];  // (component-wise select: a float4 select goes through scratch)
            qmeta[(size_t)w * maxn + k] = make_float4(valid ? m.x : 0.f, valid ? m.y : 0.f, valid ? m.z : 0.f,
                                                       valid ? m.w : __builtin_bit_cast(float, -1));
        }
        if (valid && owner) atomicMax(owner + vstart + sv, w * maxn + k);
    }
}

// FPS_TPL = largest "reference threads per lane" the register sampler is instantiated for in this
// kernel (4: lists below 512 entries, e.g. 7 x 7 x 7 windows; 8: below 1024; 16: below 2048, e.g. 11 x 11 x 11 --
// separate instantiations so that the common one keeps its register footprint)
//
// One wavefront per window.  K3 (ref :193-350) as ONE ordered hit sequence: the four query tables are scanned in
// their concatenated order (odd | even | win1_other | win2_other, 64 offsets per step); a hit appends to the
// sequence; the reference's four lists are prefixes / ranges of it (odd = the first min(#odd hits, max_odd) entries,
// even = min(#even hits, max_even) entries from #odd hits on, win1 = the first min(#hits of the first three tables,
// max_win1), win2 = the first min(#hits, max_win2)), so every hit is stored and resolved once.
//   hit test      with occupancy columns: one 64-bit word per (x, y) column of the neighbourhood (bit z), loaded
//                 once; without (z > 64): a hash probe per offset (the reference's way)
//   voxel index   sorted voxel list: column base + popcount(column word below z) -- no hash at all;
//                 any other order: one hash probe per hit, all in flight together
//
// RANKED = the voxel indices come from the column bases of a sorted level (mssvt_level_setup_sorted): that instantiation
// holds no hash probe at all -- besides the probes themselves this matters for the STORES: a load in a probe loop
// whose last result may stay unconsumed makes the compiler wait vmcnt(0) wherever its register is reused, i.e. for
// every store issued since (vmcnt retires in order: 5 full write round trips per window, 40 of 65 us of this kernel).
template <int FPS_TPL, bool RANKED>
__global__ void __launch_bounds__(PLAN_MAX_WPB *MSSVT_WAVE, FPS_TPL <= 4 ? PLAN_WAVES_PER_SIMD : 1) k_window_plan(PlanArgs a) {
    extern __shared__ int lds[];
    const int wv = threadIdx.x / MSSVT_WAVE, lane = lane_id();
    const int w = blockIdx.x * (blockDim.x / MSSVT_WAVE) + wv;
    // (issued before the window count is known: the two loads travel together; the grid is sized for the capacity)
    const int4 wi = reinterpret_cast<const int4 *>(a.win_indices)[min(w, a.win_capacity - 1)];  // [b,wz,wy,wx]
    if (w >= *a.num_wins) return;  // wave-uniform
    int si = 0;
    (void)si;
    PSTAMP()
#ifdef MSSVT_STAMPS
    if (lane == 0 && w < 32768) g_plan_span[2 * w] = __builtin_readcyclecounter();
#endif
    const int K = a.key_num_sample;
    int *base = lds + (size_t)wv * a.lds_words_per_wave;
    float4 *hmeta = reinterpret_cast<float4 *>(base);  // resolved metadata of the first meta_cap hits (16-byte aligned)
    int *hpk = base + 4 * a.meta_cap;                  // hit sequence: packed offsets in table order
    int *hsv = hpk + a.hit_cap;                        // ... and the voxel index (inside the sample) of each hit
    int *fps_out = hsv + a.hit_cap;
    unsigned long long *colw = reinterpret_cast<unsigned long long *>(fps_out + ((K + 1) & ~1));  // 8-byte aligned
    const int ncols = a.fnx * a.fny;
    int *cbase = reinterpret_cast<int *>(colw + ncols);
    float *cand = reinterpret_cast<float *>(cbase + ncols);  // 4 x tab_q words: x, y, z, (slot << 1 | valid)

    const slot_t *tab = a.table + (size_t)wi.x * a.hash_size;
    int vstart = 0;
    for (int k = 0; k < wi.x; ++k) vstart += a.v_bs_cnt[k];
    // vmcnt retires in order: a wait for a LOAD also waits for every store issued before it (a full write round trip).
    // So every load of a window is issued up here -- also voxel 0 of the sample, which an FPS-picked empty slot turns
    // into (the (x + 0.1).int() quirk below) -- and no store is issued before the column words are back.
    int4 v0 = make_int4(0, 0, 0, 0);
    if (a.kmeta1) v0 = reinterpret_cast<const int4 *>(a.indices)[vstart];  // (a live window's sample holds >= 1 voxel)
    const int cx = wi.w * a.x_ws + a.x_ws / 2, cy = wi.z * a.y_ws + a.y_ws / 2,
              cz = wi.y * a.z_ws + a.z_ws / 2;
    const float wcx = plan_centre(wi.w, a.wsx, a.minx), wcy = plan_centre(wi.z, a.wsy, a.miny),
                wcz = plan_centre(wi.y, a.wsz, a.minz);
    const int e0 = a.n_odd, e1 = e0 + a.n_even, e2 = e1 + a.n_win1, total = e2 + a.n_win2;
    const bool use_occ = RANKED || a.occ != nullptr;
    const bool ranked = RANKED;
    // (a level that turned out unsorted reports 0 windows; should a caller pass window rows of its own: nothing to do)
    if (RANKED && (a.level_status[0] & ST_UNSORTED)) return;
    int pre[PLAN_PRE];
    if (use_occ) {
        // the first PLAN_PRE x 64 offsets travel together with the column words
        // (only the steps that will run: a load whose result is never consumed stays "pending" for the compiler, which
        // then waits vmcnt(0) -- i.e. for every store issued since -- wherever its register is reused)
#pragma unroll
        for (int ci = 0; ci < PLAN_PRE; ++ci) {
            const int q = ci * MSSVT_WAVE + lane;
            pre[ci] = PACK0;
            if (ci * MSSVT_WAVE < total) pre[ci] = a.q_packed[min(q, total - 1)];
        }
        for (int c = lane; c < ncols; c += MSSVT_WAVE) {
            const int ccx = (int)(((unsigned int)c * (unsigned int)a.fny_magic) >> 20);  // = c / fny (checked by the host)
            const int sx = cx + a.fx0 + ccx, sy = cy + a.fy0 + (c - ccx * a.fny);
            const bool in = sx >= 0 && sx < a.x_max && sy >= 0 && sy < a.y_max;
            const size_t col = ((size_t)wi.x * a.x_max + (in ? sx : 0)) * a.y_max + (in ? sy : 0);
            const unsigned long long word = a.occ[col];
            colw[c] = in ? word : 0ull;
            if (ranked) cbase[c] = a.col_vbase[col];
        }
        wave_lds_sync();
    }
    // metadata of the sample's voxel 0 (see above; consumed here so that no load is pending once the stores start)
    const float4 m0 = make_float4(plan_centre(v0.w, a.vsx, a.minx) - wcx, plan_centre(v0.z, a.vsy, a.miny) - wcy,
                                  plan_centre(v0.y, a.vsz, a.minz) - wcz, __builtin_bit_cast(float, vstart));
    PSTAMP()
    // ---- K3: the hit sequence ---------------------------------------------------------------------------
    // hits of the first 1 / 2 / 3 / 4 tables; a count is fixed in the step that holds its table's end (-1: not reached,
    // every hit so far belongs to it)
    int cnt_odd = -1, cnt_le1 = -1, cnt_le2 = -1, cnt_all = 0;
    bool stop = false;
    // one 64-offset step: `hit` / `pk` / `sv` of this lane's offset -> the sequence and the counts
    auto append = [&](int bq, bool hit, int pk, int sv) __attribute__((always_inline)) {
        const unsigned long long m_all = __ballot(hit);
        // position of a hit in this lane = hits before it; the hits before lane (e - bq) are those of the tables below e
        const int p = cnt_all + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m_all >> 32),
                                                             __builtin_amdgcn_mbcnt_lo((unsigned int)m_all, 0u));
        // (branch-free: the steps of the unrolled part stay one basic block and their LDS reads overlap)
        const int r0 = __builtin_amdgcn_readlane(p, (e0 - bq) & 63), r1 = __builtin_amdgcn_readlane(p, (e1 - bq) & 63),
                  r2 = __builtin_amdgcn_readlane(p, (e2 - bq) & 63);
        cnt_odd = (unsigned int)(e0 - bq) < 64u ? r0 : cnt_odd;
        cnt_le1 = (unsigned int)(e1 - bq) < 64u ? r1 : cnt_le1;
        cnt_le2 = (unsigned int)(e2 - bq) < 64u ? r2 : cnt_le2;
        if (hit) {
            hpk[p] = pk;
            if (!use_occ) hsv[p] = sv;
        }
        cnt_all += __popcll(m_all);
        // no list takes another entry (ref :238, :274, :310)?
        const int nb = bq + MSSVT_WAVE;
        stop = cnt_all >= a.max_win2 && (nb >= e2 || cnt_all >= a.max_win1) &&
               (nb >= e1 || (nb >= e0 && cnt_all - cnt_odd >= a.max_even)) && (nb >= e0 || cnt_all >= a.max_odd);
    };
    if (use_occ) {
        // the offset's column word sits at colw[pk >> 21]; columns outside the grid are zero words, z outside
        // [0, z_max) has no bit set
        auto occ_step = [&](int bq, int pk) __attribute__((always_inline)) {
            const int sz = cz + ((pk >> 14) & 127) - 64;
            bool hit = false;
            if ((unsigned int)sz < (unsigned int)a.z_max) hit = (colw[(unsigned int)pk >> 21] >> sz) & 1ull;  // (PACK0 lanes: q >= total)
            append(bq, hit && bq + lane < total, pk, MSSVT_EMPTY);
        };
        // (the preloaded steps run without the early stop: their LDS reads are independent and overlap; the lists are
        // cut to their capacity below either way)
#pragma unroll
        for (int ci = 0; ci < PLAN_PRE; ++ci) occ_step(ci * MSSVT_WAVE, pre[ci]);  // (a step past the tables finds no hit)
        for (int bq = PLAN_PRE * MSSVT_WAVE; bq < total && !stop; bq += MSSVT_WAVE)
            occ_step(bq, bq + lane < total ? a.q_packed[bq + lane] : PACK0);
    } else if (!RANKED) {
        for (int bq = 0; bq < total && !stop; bq += MSSVT_WAVE) {
            const int q = bq + lane;
            int pk = PACK0, sv = MSSVT_EMPTY;
            if (q < total) {
                const int seg = (q >= e0) + (q >= e1) + (q >= e2);
                const int *src = seg == 0 ? a.q_odd + q * 3
                               : seg == 1 ? a.q_even + (q - e0) * 3
                               : seg == 2 ? a.q_win1 + (q - e1) * 3
                                          : a.q_win2 + (q - e2) * 3;
                const int ox = src[0], oy = src[1], oz = src[2];
                pk = pack_off(ox, oy, oz);
                const int sx = cx + ox, sy = cy + oy, sz = cz + oz;
                if (!(sx >= a.x_max || sx < 0 || sy >= a.y_max || sy < 0 || sz >= a.z_max || sz < 0))
                    sv = table_find(sx * a.y_max * a.z_max + sy * a.z_max + sz, a.hash_size, tab);
            }
            append(bq, sv != MSSVT_EMPTY, pk, sv);
        }
    }
    if (cnt_odd < 0) cnt_odd = cnt_all;
    if (cnt_le1 < 0) cnt_le1 = cnt_all;
    if (cnt_le2 < 0) cnt_le2 = cnt_all;
    wave_lds_sync();
    const int nO = min(cnt_odd, a.max_odd), nE = min(cnt_le1 - cnt_odd, a.max_even), n1 = min(cnt_le2, a.max_win1),
              n2 = min(cnt_all, a.max_win2);
    {
        // voxel index of every hit some list holds (with columns), resolved metadata of those the odd / even / win1
        // lists and the win1 keys hold
        const int mneed = a.kmeta1 ? min(max(n1, max(nO, cnt_odd + nE)), a.meta_cap) : 0;
        const int hmax = max(max(n1, n2), max(nO, cnt_odd + nE));
        for (int e = lane; e < hmax; e += MSSVT_WAVE) {
            const int pk = hpk[e];
            int ox, oy, oz;
            fps_unpack(pk, ox, oy, oz);
            int sv;
            if (RANKED) {
                const int c = (unsigned int)pk >> 21;
                sv = cbase[c] + __popcll(colw[c] & ((1ull << (cz + oz)) - 1ull));
                hsv[e] = sv;
            } else if (!use_occ) {
                sv = hsv[e];
            } else {
                const int sz = cz + oz;
                {
                    sv = a.table ? table_find((cx + ox) * a.y_max * a.z_max + (cy + oy) * a.z_max + sz, a.hash_size, tab) : MSSVT_EMPTY;
                    // an occupied cell the table does not know: only after a voxel-table overflow (status bit set, the
                    // caller raises at the end of the frame).  Until then every index downstream must stay in range and
                    // the counts must match the entries: the entry is redirected to the sample's first voxel.
                    if (sv == MSSVT_EMPTY) sv = 0;
                }
                hsv[e] = sv;
            }
            if (e < mneed)
                hmeta[e] = make_float4(plan_centre(cx + ox, a.vsx, a.minx) - wcx, plan_centre(cy + oy, a.vsy, a.miny) - wcy,
                                       plan_centre(cz + oz, a.vsz, a.minz) - wcz, __builtin_bit_cast(float, vstart + sv));
        }
        wave_lds_sync();
        // every load is back before the first store goes out -- and the compiler knows it: a load it believes pending
        // on some path (the probe loop, a table step beyond the preloaded ones) would otherwise cost a vmcnt(0) after stores
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    PSTAMP()
    plan_emit_list(w, lane, a.max_odd, nO, 0, hsv, hmeta, a.ind_odd, a.kmeta1 ? a.qmeta_odd : nullptr, a.owner_odd, vstart);
    plan_emit_list(w, lane, a.max_even, nE, cnt_odd, hsv, hmeta, a.ind_even, a.kmeta1 ? a.qmeta_even : nullptr, a.owner_even, vstart);
    plan_emit_list(w, lane, a.max_win1, n1, 0, hsv, hmeta, a.ind_win1, a.kmeta1 ? a.qmeta_win1 : nullptr, a.owner_win1, vstart);
    if (lane == 0) {
        a.win_vstart[w] = vstart;
        if (a.kmeta1) {
            a.wcentre[w] = make_float4(wcx, wcy, wcz, 0.f);
            a.nq_valid[w] = nO;
            a.nq_valid[a.win_capacity + w] = nE;
            a.nq_valid[2 * a.win_capacity + w] = n1;
        }
    }

    // ---- interpolation tables (see PlanArgs::tabs) ----------------------------------------------------------
    // Lanes are dealt to (table, win1 slot) pairs: with n1 <= 32 entries two tables go through the 3-NN search side by
    // side (lanes 0-31 / 32-63), with n1 <= 16 four; the candidates of table t sit in cand + 4 t tab_q.
    if (a.n_tabs > 0) {
        const int per = n1 <= 16 ? 16 : n1 <= 32 ? 32 : 64, side = MSSVT_WAVE / per;  // lanes per table, tables side by side
        for (int t0 = 0; t0 < a.n_tabs; t0 += side) {
            // candidates of tables t0 .. t0 + side - 1: known points = ALL query slots; empty slots sit at the world origin
            // with zero features (ref :302).  In slot order: every valid slot (they come first), and of the EMPTY slots
            // only the first three -- all of them are the same point, the search keeps the first seen on ties (strict <):
            // a fourth can never enter the best three
            int nc_mine = 0, first_mine = 0, maxn_mine = 1, zero_mine = 0, interp_mine = 0, nvl_mine = 0;
            int4 *row_mine = nullptr;
            float4 *w_mine = nullptr;
            const int tl = lane / per;  // this lane's table (relative to t0)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                if (ti < t0 || ti >= t0 + side || ti >= a.n_tabs) continue;
                const int first = a.tabs[ti].list == 1 ? cnt_odd : 0;
                const int nvl = a.tabs[ti].list == 0 ? nO : a.tabs[ti].list == 1 ? nE : n1, maxn = a.tabs[ti].maxn;
                const int ncand = a.tabs[ti].interp ? nvl + min(3, maxn - nvl) : 0;
                // one 16-byte entry per candidate: (x, y, z, slot << 1 | valid) -- one LDS read per candidate in the search below
                float4 *kc = reinterpret_cast<float4 *>(cand) + (ti - t0) * a.tab_q;
                for (int k = lane; k < ncand; k += MSSVT_WAVE) {
                    float x = 0.f, y = 0.f, z = 0.f;
                    if (k < nvl) {
                        int ox, oy, oz;
                        fps_unpack(hpk[first + k], ox, oy, oz);
                        x = plan_centre(cx + ox, a.vsx, a.minx);
                        y = plan_centre(cy + oy, a.vsy, a.miny);
                        z = plan_centre(cz + oz, a.vsz, a.minz);
                    }
                    kc[k] = make_float4(x, y, z, __builtin_bit_cast(float, (k << 1) | (k < nvl ? 1 : 0)));  // slot, valid bit
                }
                if (tl == ti - t0) {
                    nc_mine = ncand; first_mine = first; maxn_mine = maxn; zero_mine = a.tabs[ti].zero_row;
                    interp_mine = a.tabs[ti].interp; nvl_mine = nvl;
                    row_mine = a.tabs[ti].tab_row; w_mine = a.tabs[ti].tab_w;
                }
            }
            wave_lds_sync();
            const float4 *kc = reinterpret_cast<const float4 *>(cand) + tl * a.tab_q;
            if (row_mine != nullptr && !interp_mine) {  // ref mssvt_backbone.py:327-330: only the query voxels are updated
                for (int k = lane % per; k < nvl_mine; k += per) {
                    const int v = hsv[first_mine + k];
                    row_mine[vstart + v] = make_int4(w * maxn_mine + k, zero_mine, zero_mine, 0);
                    w_mine[vstart + v] = make_float4(1.f, 0.f, 0.f, 0.f);
                }
            } else if (row_mine != nullptr) {
                for (int sl = lane % per; sl < n1; sl += per) {  // K9 (ref interpolate_gpu.cu:16-59) + weights (ref :305-307)
                    const int v = hsv[sl];
                    int ox, oy, oz;
                    fps_unpack(hpk[sl], ox, oy, oz);
                    const float ux = plan_centre(cx + ox, a.vsx, a.minx), uy = plan_centre(cy + oy, a.vsy, a.miny),
                                uz = plan_centre(cz + oz, a.vsz, a.minz);
                    // the reference keeps the running bests in double (initial 1e40) and compares float distances against
                    // them: the same order as float compares against +inf
                    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
                    int c1 = -1, c2 = -1, c3 = -1;
                    for (int k = 0; k < nc_mine; ++k) {
                        const float4 kp = kc[k];
                        const float dx = ux - kp.x, dy = uy - kp.y, dz = uz - kp.z;
                        const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                        if (d < b1) { b3 = b2; c3 = c2; b2 = b1; c2 = c1; b1 = d; c1 = k; }
                        else if (d < b2) { b3 = b2; c3 = c2; b2 = d; c2 = k; }
                        else if (d < b3) { b3 = d; c3 = k; }
                    }
                    // fewer than three candidates (nq < 3): the reference leaves index 0 / distance 1e40 -> weight ~0
                    const int m1 = c1 >= 0 ? __builtin_bit_cast(int, kc[c1].w) : 0, m2 = c2 >= 0 ? __builtin_bit_cast(int, kc[c2].w) : 0,
                              m3 = c3 >= 0 ? __builtin_bit_cast(int, kc[c3].w) : 0;
                    // (hardware square root / reciprocal, 1 ulp each: the IEEE-exact library forms are ~85 instructions per voxel
                    // for weights whose consumers carry an fp32 tolerance; the reference itself mixes float and double here)
                    const float d1 = fmaxf(c1 >= 0 ? __builtin_amdgcn_sqrtf(b1) : INFINITY, 1e-10f),
                                d2 = fmaxf(c2 >= 0 ? __builtin_amdgcn_sqrtf(b2) : INFINITY, 1e-10f),
                                d3 = fmaxf(c3 >= 0 ? __builtin_amdgcn_sqrtf(b3) : INFINITY, 1e-10f);
                    float w1 = __builtin_amdgcn_rcpf(d1), w2 = __builtin_amdgcn_rcpf(d2), w3 = __builtin_amdgcn_rcpf(d3);
                    const float rnorm = __builtin_amdgcn_rcpf((w1 + w2) + w3);
                    w1 *= rnorm; w2 *= rnorm; w3 *= rnorm;
                    if (!(m1 & 1) || c1 < 0) w1 = 0.f;  // empty slots carry zero features
                    if (!(m2 & 1) || c2 < 0) w2 = 0.f;
                    if (!(m3 & 1) || c3 < 0) w3 = 0.f;
                    row_mine[vstart + v] = make_int4(w1 != 0.f ? w * maxn_mine + (m1 >> 1) : zero_mine,
                                                     w2 != 0.f ? w * maxn_mine + (m2 >> 1) : zero_mine,
                                                     w3 != 0.f ? w * maxn_mine + (m3 >> 1) : zero_mine, 0);
                    w_mine[vstart + v] = make_float4(w1, w2, w3, 0.f);
                }
            }
            wave_lds_sync();
        }
    }
    PSTAMP()
    // ---- K7 + K8 + masks for both scales (ref mssvt_backbone.py:247-258) -----------
    const float4 none = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, -1));
    for (int scale = 0; scale < 2; ++scale) {
        const int n = scale ? a.max_win2 : a.max_win1;
        const int bs = scale ? a.bs2 : a.bs1;
        const int nv = scale ? n2 : n1;
        if (nv <= MSSVT_WAVE && nv <= bs && bs >= 2) {
            const int pk0 = lane < nv ? hpk[lane] : PACK0;
            if (K > MSSVT_WAVE)
                fps_on_list_fast<false, 4>(pk0, n, nv, K, bs, fps_out, lane);
            else if (nv <= 16)
                fps_on_list_fast<true, 1>(pk0, n, nv, K, bs, fps_out, lane);
            else if (nv <= 32)
                fps_on_list_fast<true, 2>(pk0, n, nv, K, bs, fps_out, lane);
            else
                fps_on_list_fast<true, 4>(pk0, n, nv, K, bs, fps_out, lane);
        } else if (bs <= 64)
            fps_on_list_regs<1>(hpk, nv, n, K, bs, fps_out, lane);
        else if (bs == 128)
            fps_on_list_regs<2>(hpk, nv, n, K, bs, fps_out, lane);
        else if (bs == 256)
            fps_on_list_regs<4>(hpk, nv, n, K, bs, fps_out, lane);
        else if (FPS_TPL >= 8 && bs == 512)
            fps_on_list_regs<FPS_TPL >= 8 ? 8 : 1>(hpk, nv, n, K, bs, fps_out, lane);
        else if (FPS_TPL >= 16 && bs == 1024)
            fps_on_list_regs<FPS_TPL >= 16 ? 16 : 1>(hpk, nv, n, K, bs, fps_out, lane);
        wave_lds_sync();
        PSTAMP()
        int *kout0 = scale ? a.k_ind2 : a.k_ind1;
        unsigned char *mout0 = scale ? a.k_mask2 : a.k_mask1;
        int *kout = kout0 ? kout0 + (size_t)w * K : nullptr;
        unsigned char *mout = mout0 ? mout0 + (size_t)w * K : nullptr;
        for (int j = lane; j < K; j += MSSVT_WAVE) {
            const int f = fps_out[j];
            const bool entry = f < nv;  // a list entry (else an empty slot: index -1, offset (0,0,0))
            // ref :253-256: the index is round-tripped through fp32 and "(x + 0.1).int()"
            // truncates toward zero -> a picked EMPTY slot (-1) becomes voxel 0 of the sample
            const int kid = (int)((float)(entry ? hsv[f] : MSSVT_EMPTY) + 0.1f);
            if (kout) kout[j] = kid;
            const bool masked = (j > 0 && f == 0) || kid < 0;
            if (mout) mout[j] = (unsigned char)(masked ? 1 : 0);
            if (a.kmeta1) {
                float4 m = none;
                if (!masked) {
                    if (entry && f < a.meta_cap && scale == 0) {
                        m = hmeta[f];  // (the win1 keys are win1 list entries: resolved above)
                    } else {
                        if (entry) {  // a list entry: its voxel is the window centre cell + offset
                            int ox, oy, oz;
                            fps_unpack(hpk[f], ox, oy, oz);
                            const float px = plan_centre(cx + ox, a.vsx, a.minx), py = plan_centre(cy + oy, a.vsy, a.miny),
                                        pz = plan_centre(cz + oz, a.vsz, a.minz);
                            m = make_float4(px - wcx, py - wcy, pz - wcz, __builtin_bit_cast(float, vstart + kid));
                        } else {  // the reference quirk: an empty slot became voxel 0 of the sample (kid == 0)
                            m = m0;
                        }
                    }
                }
                (scale ? a.kmeta2 : a.kmeta1)[(size_t)w * K + j] = m;
            }
        }
        wave_lds_sync();
        PSTAMP()
    }
#ifdef MSSVT_STAMPS
    if (lane == 0 && w < 32768) g_plan_span[2 * w + 1] = __builtin_readcyclecounter();
#endif
}

static inline int plan_opt_n_threads(int work_size) {  // ref cuda_utils.h:10-14
    const int pow_2 = (int)(log((double)work_size) / log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

extern "C" int mssvt_window_plan_two(
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
    const int *host_tab_zero_row, int *const *host_tab_row, float *const *host_tab_w, void *stream) {
    // (ind_* / k_ind* / k_mask* / owner_*: each optional when the resolved metadata is asked for -- the fused consumers read
    // kmeta / qmeta / the tables; without metadata they are the call's only outputs)
    const bool lists_wanted = ind_odd && ind_even && ind_win1 && k_ind1 && k_ind2 && k_mask1 && k_mask2 && owner_win1 && owner_odd && owner_even;
    if (!win_indices || !num_wins_dev || !v_bs_cnt || !win_vstart || (!kmeta1 && !lists_wanted) || hash_size <= 0 || key_num_sample <= 0 || max_num_win1 <= 0 ||
        max_num_win2 <= 0 || max_num_odd <= 0 || max_num_even <= 0 || (column_vbase && !level_status_dev))
        return MSSVT_E_BADARG;
    if (win_capacity <= 0) return MSSVT_OK;
    // offsets are packed into bytes (|offset| <= 60) -- far beyond any window in use
    if (x_ws > 60 || y_ws > 60 || z_ws > 60 || key_num_sample > 1024) return MSSVT_E_TOOLARGE;
    // the register samplers examine slots [0, 2 bs) with bs = min(2^floor(log2 n), 1024) (the reference's block
    // size): lists of 2048 slots or more (e.g. 13 x 13 x 13 windows) are not covered -> the caller's operator path
    if (max_num_win1 >= 2048 || max_num_win2 >= 2048) return MSSVT_E_TOOLARGE;
    PlanArgs a;
    a.x_max = x_max; a.y_max = y_max; a.z_max = z_max;
    a.x_ws = x_ws; a.y_ws = y_ws; a.z_ws = z_ws;
    a.max_odd = max_num_odd; a.max_even = max_num_even; a.max_win1 = max_num_win1; a.max_win2 = max_num_win2;
    a.hash_size = hash_size; a.batch_size = batch_size;
    a.n_odd = num_odd; a.n_even = num_even; a.n_win1 = num_win1; a.n_win2 = num_win2;
    a.q_odd = vox_query_odd; a.q_even = vox_query_even; a.q_win1 = vox_query_win1; a.q_win2 = vox_query_win2;
    a.key_num_sample = key_num_sample;
    a.bs1 = plan_opt_n_threads(max_num_win1);
    a.bs2 = plan_opt_n_threads(max_num_win2);
    a.win_indices = win_indices; a.num_wins = num_wins_dev;
    a.table = reinterpret_cast<const slot_t *>(xyz_to_vidx);
    a.v_bs_cnt = v_bs_cnt;
    a.ind_odd = ind_odd; a.ind_even = ind_even; a.ind_win1 = ind_win1;
    a.k_ind1 = k_ind1; a.k_ind2 = k_ind2; a.k_mask1 = k_mask1; a.k_mask2 = k_mask2;
    a.win_vstart = win_vstart;
    a.owner_win1 = owner_win1; a.owner_odd = owner_odd; a.owner_even = owner_even;
    a.qmeta_odd = reinterpret_cast<float4 *>(qmeta_odd);
    a.qmeta_even = reinterpret_cast<float4 *>(qmeta_even);
    a.qmeta_win1 = reinterpret_cast<float4 *>(qmeta_win1);
    a.kmeta1 = reinterpret_cast<float4 *>(kmeta1);
    a.kmeta2 = reinterpret_cast<float4 *>(kmeta2);
    a.wcentre = reinterpret_cast<float4 *>(wcentre);
    a.nq_valid = nq_valid;
    a.win_capacity = win_capacity;
    a.indices = indices;
    if (kmeta1) {
        if (!kmeta2 || !wcentre || !nq_valid || !indices || !host_voxel_size3 || !host_range_min3 || !host_win_size3)
            return MSSVT_E_BADARG;  // (qmeta_odd / _even / _win1: each optional)
        a.vsx = host_voxel_size3[0]; a.vsy = host_voxel_size3[1]; a.vsz = host_voxel_size3[2];
        a.minx = host_range_min3[0]; a.miny = host_range_min3[1]; a.minz = host_range_min3[2];
        a.wsx = host_win_size3[0]; a.wsy = host_win_size3[1]; a.wsz = host_win_size3[2];
    } else {
        a.vsx = a.vsy = a.vsz = a.minx = a.miny = a.minz = a.wsx = a.wsy = a.wsz = 0.f;
    }
    const int bsmax = a.bs1 > a.bs2 ? a.bs1 : a.bs2;
    const int total = num_odd + num_even + num_win1 + num_win2;
    a.hit_cap = (total + 1) & ~1;
    a.meta_cap = 0;
    if (kmeta1) {
        a.meta_cap = max_num_win1 > num_odd + max_num_even ? max_num_win1 : num_odd + max_num_even;
        if (a.meta_cap > a.hit_cap) a.meta_cap = a.hit_cap;
    }
    a.lds_words_per_wave = 4 * a.meta_cap + 2 * a.hit_cap + ((key_num_sample + 1) & ~1);
    a.occ = nullptr;
    a.col_vbase = a.level_status = nullptr;
    a.win_counts = win_counts_dev;
    a.n_tabs = 0;
    a.tab_q = 0;
    if (num_tabs < 0 || num_tabs > 4) return MSSVT_E_TOOLARGE;
    if (num_tabs > 0) {
        if (!kmeta1 || !host_tab_list || !host_tab_interp || !host_tab_zero_row || !host_tab_row || !host_tab_w) return MSSVT_E_BADARG;
        for (int t = 0; t < num_tabs; ++t) {
            if (host_tab_list[t] < 0 || host_tab_list[t] > 2 || !host_tab_row[t] || !host_tab_w[t]) return MSSVT_E_BADARG;
            a.tabs[t].list = host_tab_list[t];
            a.tabs[t].maxn = host_tab_list[t] == 0 ? max_num_odd : host_tab_list[t] == 1 ? max_num_even : max_num_win1;
            a.tabs[t].interp = host_tab_interp[t];
            a.tabs[t].zero_row = host_tab_zero_row[t];
            a.tabs[t].tab_row = reinterpret_cast<int4 *>(host_tab_row[t]);
            a.tabs[t].tab_w = reinterpret_cast<float4 *>(host_tab_w[t]);
            if (a.tabs[t].maxn + 3 > a.tab_q) a.tab_q = a.tabs[t].maxn + 3;
        }
        a.tab_q = (a.tab_q + 1) & ~1;
        a.n_tabs = num_tabs;
    }
    a.fx0 = a.fy0 = a.fnx = a.fny = a.fny_magic = 0;
    a.q_packed = packed_offsets;
    if (occ_columns && host_footprint4 && packed_offsets && z_max <= 64 && host_footprint4[2] > 0 &&
        host_footprint4[3] > 0 && host_footprint4[2] * host_footprint4[3] <= 1024) {
        a.occ = occ_columns;
        a.fx0 = host_footprint4[0]; a.fy0 = host_footprint4[1];
        a.fnx = host_footprint4[2]; a.fny = host_footprint4[3];
        a.fny_magic = (1 << 20) / a.fny + 1;
        for (int c = 0; c < a.fnx * a.fny; ++c)
            if ((int)(((unsigned int)c * (unsigned int)a.fny_magic) >> 20) != c / a.fny) return MSSVT_E_TOOLARGE;  // (never: c < 1024)
        a.col_vbase = column_vbase;
        a.level_status = level_status_dev;
        // + the column words (8 bytes each) and the column bases
        a.lds_words_per_wave += 3 * a.fnx * a.fny;
        a.lds_words_per_wave += a.lds_words_per_wave & 1;  // keep every wave's region 8-byte aligned
    } else if (!xyz_to_vidx || !vox_query_odd || !vox_query_even || !vox_query_win1 || !vox_query_win2) {
        return MSSVT_E_BADARG;
    }
    if (!a.col_vbase && !xyz_to_vidx) return MSSVT_E_BADARG;  // the hash is the only source of voxel indices then
    a.lds_words_per_wave += 4 * a.tab_q * 4;  // candidates of up to four tables side by side
    a.lds_words_per_wave += a.lds_words_per_wave & 1;
    // waves (windows) per workgroup: the count that puts the most waves on a CU (160 KiB of LDS,
    // workgroups <= 64 KiB), ties -> larger workgroups (fewer of them to dispatch)
    int wpb = 1, best_waves = 0;
    for (int cand = 1; cand <= PLAN_MAX_WPB; ++cand) {
        const size_t bytes = (size_t)a.lds_words_per_wave * 4 * cand;
        if (bytes > 64 * 1024) break;
        const int waves = (int)((160 * 1024) / bytes) * cand;
        if (waves >= best_waves) {
            best_waves = waves;
            wpb = cand;
        }
    }
    if (getenv("MSSVT_PLAN_WPB")) wpb = atoi(getenv("MSSVT_PLAN_WPB"));
    const size_t lds_bytes = (size_t)a.lds_words_per_wave * 4 * wpb;
    if (lds_bytes > 160 * 1024) return MSSVT_E_TOOLARGE;
    // lists of 512 .. 1023 / 1024 .. 2047 slots: the instantiations with the 8- / 16-slot register samplers
    const int tpl = bsmax > 512 ? 16 : bsmax > 256 ? 8 : 4;
    const bool ranked = a.col_vbase != nullptr;
    typedef void (*plan_kernel_t)(PlanArgs);
    const plan_kernel_t kernel = ranked ? (tpl == 16 ? k_window_plan<16, true> : tpl == 8 ? k_window_plan<8, true> : k_window_plan<4, true>)
                                        : (tpl == 16 ? k_window_plan<16, false> : tpl == 8 ? k_window_plan<8, false> : k_window_plan<4, false>);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    const dim3 grid(divup(win_capacity, wpb)), block(wpb * MSSVT_WAVE);
    hipStream_t st = (hipStream_t)stream;
    kernel<<<grid, block, lds_bytes, st>>>(a);
    return mssvt_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Occupancy columns of a voxel set: one 64-bit word per (b, x, y), bit z = the cell holds a voxel
// (z_max <= 64).  1.8 MB for a 470 x 470 grid: L2 resident.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_occupancy_columns(const int *indices, int n, int batch_size, int x_max,
                                                           int y_max, int z_max, unsigned long long *cols) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 v = reinterpret_cast<const int4 *>(indices)[i];  // [b, z, y, x]
    if (v.x < 0 || v.x >= batch_size || v.w < 0 || v.w >= x_max || v.z < 0 || v.z >= y_max || v.y < 0 || v.y >= z_max)
        return;
    atomicOr(cols + ((size_t)v.x * x_max + v.w) * y_max + v.z, 1ull << v.y);
}

extern "C" int mssvt_occupancy_columns(const int *indices, int num_voxels, int batch_size, int x_max, int y_max,
                                       int z_max, unsigned long long *columns, void *stream_) {
    if (!columns || (!indices && num_voxels > 0) || num_voxels < 0 || batch_size <= 0 || x_max <= 0 || y_max <= 0 ||
        z_max <= 0)
        return MSSVT_E_BADARG;
    if (z_max > 64) return MSSVT_E_TOOLARGE;
    hipStream_t stream = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(columns, 0, (size_t)batch_size * x_max * y_max * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return (int)e;
    return mssvt_occupancy_columns_launch(indices, num_voxels, batch_size, x_max, y_max, z_max, columns, stream);
}

int mssvt_occupancy_columns_launch(const int *indices, int num_voxels, int batch_size, int x_max, int y_max, int z_max,
                                   unsigned long long *columns, hipStream_t stream) {
    if (num_voxels > 0)
        k_occupancy_columns<<<divup(num_voxels, 256), 256, 0, stream>>>(indices, num_voxels, batch_size, x_max, y_max,
                                                                      z_max, columns);
    return mssvt_launch_status();
}

// ---------------------------------------------------------------------------------------------
// Work order for the attention kernel: windows sorted by DESCENDING number of valid queries
// (their cost varies from 0 to 20+ queries; heavy ones first keeps the tail of the persistent
// kernel short -- longest-processing-time-first), windows without a query dropped.
// One workgroup: LDS histogram -> scan -> scatter (counting sort; order inside a bucket is
// arbitrary, which is harmless: every window writes only its own rows).
// ---------------------------------------------------------------------------------------------
#define PO_WAVES 16
#define PO_KEYS 257
#define PO_V 8  // window batches of 64 in flight per wave (one global-load latency per PO_V batches)
#define PO_MAX_SETS 4
struct PlanOrderSet {
    const int *nq_valid;
    int max_key, nq;
    const float4 *qmeta;
    int *perm, *num_active, *q_off, *num_rows;
    float4 *rmeta;
    int2 *rsrc;
};
struct PlanOrderPack {
    PlanOrderSet s[PO_MAX_SETS];
};

// blockIdx.y = query list (the lists of a plan are ordered in one launch), blockIdx.x = one of G workgroups that
// each own a contiguous run of windows.  G > 1: PHASE 0 leaves every workgroup's key histogram and row total in
// `scratch` (G x (PO_KEYS + 1) ints, the head of the list's row array that k_query_rows only fills afterwards),
// PHASE 1 turns the other workgroups' histograms into its offsets and then sorts its own run exactly like the
// single workgroup does.  G == 1: PHASE 1 alone.
template <int PHASE>
__global__ void __launch_bounds__(PO_WAVES *MSSVT_WAVE) k_plan_order(const int *num_wins, int row_capacity,
                                                                     PlanOrderPack pack) {
    const PlanOrderSet &ps = pack.s[blockIdx.y];
    const int *nq_valid = ps.nq_valid;
    const int max_key = ps.max_key;
    const int G = gridDim.x, gidx = blockIdx.x;
    int *scratch = reinterpret_cast<int *>(ps.rsrc);
    int *perm = ps.perm, *num_active = ps.num_active, *q_off = ps.q_off, *num_rows = ps.num_rows;
    // per-wave histograms: copies sit PO_KEYS (odd) words apart -> different LDS banks, so the 16
    // waves' atomics on the few populated keys proceed in parallel
    __shared__ int hist[PO_WAVES][PO_KEYS];
    __shared__ int key_base[PO_KEYS];
    __shared__ int wave_q[PO_WAVES];
    __shared__ int rows_before, rows_all;
    const int nw = *num_wins;
    const int wv = threadIdx.x / MSSVT_WAVE, lane = lane_id();
    for (int k = threadIdx.x; k < PO_WAVES * PO_KEYS; k += blockDim.x) (&hist[0][0])[k] = 0;
    __syncthreads();
    // each workgroup, and inside it each wave, owns a contiguous run of windows
    const int per_wg = (nw + G - 1) / G;
    const int gb = min(gidx * per_wg, nw), ge = min(gb + per_wg, nw);
    const int per_wave = (ge - gb + PO_WAVES - 1) / PO_WAVES;
    const int wb = min(gb + wv * per_wave, ge), we = min(wb + per_wave, ge);
    // pass 0: histogram of the query counts + the run's total
    int total = 0;
    for (int base = wb; base < we; base += MSSVT_WAVE * PO_V) {
        int v[PO_V];
#pragma unroll
        for (int u = 0; u < PO_V; ++u) {
            const int w = base + u * MSSVT_WAVE + lane;
            v[u] = w < we ? nq_valid[w] : -1;
        }
#pragma unroll
        for (int u = 0; u < PO_V; ++u)
            if (v[u] >= 0) {
                atomicAdd(&hist[wv][min(v[u], max_key)], 1);
                total += v[u];
            }
    }
    total = wave_sum_i(total);
    if (lane == 0) wave_q[wv] = total;
    __syncthreads();
    if (PHASE == 0) {  // publish this workgroup's histogram and row total
        if (threadIdx.x <= max_key) {
            int c = 0;
            for (int i = 0; i < PO_WAVES; ++i) c += hist[i][threadIdx.x];
            scratch[gidx * (PO_KEYS + 1) + threadIdx.x] = c;
        }
        if (threadIdx.x == 0) {
            int rows = 0;
            for (int i = 0; i < PO_WAVES; ++i) rows += wave_q[i];
            scratch[gidx * (PO_KEYS + 1) + PO_KEYS] = rows;
        }
        return;
    }
    // hist[wv][k] -> exclusive prefix over the earlier workgroups and the waves, key totals -> key_base
    // (heaviest key first)
    if (threadIdx.x <= max_key) {
        int run = 0, all = 0;
        // (8 requests in flight: one by one the other workgroups' counts were G <= 64 dependent L2 round trips, most of this launch)
        for (int g0 = 0; g0 < G && G > 1; g0 += 8) {
            int c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = g0 + u < G ? scratch[(g0 + u) * (PO_KEYS + 1) + threadIdx.x] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                run += g0 + u < gidx ? c[u] : 0;
                all += c[u];
            }
        }
        int own = 0;
        for (int i = 0; i < PO_WAVES; ++i) {
            const int c = hist[i][threadIdx.x];
            hist[i][threadIdx.x] = run;
            run += c;
            own += c;
        }
        key_base[threadIdx.x] = G > 1 ? all : own;
    }
    if (threadIdx.x == PO_WAVES * MSSVT_WAVE - 1) {
        int before = 0, all = 0;
        for (int g0 = 0; g0 < G && G > 1; g0 += 8) {
            int c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = g0 + u < G ? scratch[(g0 + u) * (PO_KEYS + 1) + PO_KEYS] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                before += g0 + u < gidx ? c[u] : 0;
                all += c[u];
            }
        }
        if (G == 1)
            for (int i = 0; i < PO_WAVES; ++i) all += wave_q[i];
        rows_before = before;
        rows_all = all;
    }
    __syncthreads();
    if (wv == 0) {
        // key_base[k] = windows with more queries than k (heaviest key first): a descending exclusive prefix, 4 keys per
        // lane + one wave scan instead of one thread walking up to 256 keys (PO_KEYS <= 4 x 64 + 1; key 0 -- windows
        // without a query -- is not placed)
        int c[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = max_key - (4 * lane + j);
            c[j] = k >= 1 ? key_base[k] : 0;
            sum += c[j];
        }
        int incl = sum;
        for (int off = 1; off < MSSVT_WAVE; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        int run = incl - sum;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = max_key - (4 * lane + j);
            if (k >= 1) key_base[k] = run;
            run += c[j];
        }
        if (gidx == 0 && lane == MSSVT_WAVE - 1) {
            *num_active = incl;
            *num_rows = rows_all < row_capacity ? rows_all : row_capacity;  // rows past the capacity are dropped (caller's bound)
        }
    }
    __syncthreads();
    // pass 1: q_off = exclusive scan of nq_valid in window order; perm = counting sort
    int carry = rows_before;
    for (int i = 0; i < wv; ++i) carry += wave_q[i];
    for (int base = wb; base < we; base += MSSVT_WAVE * PO_V) {
        int v[PO_V];
#pragma unroll
        for (int u = 0; u < PO_V; ++u) {
            const int w = base + u * MSSVT_WAVE + lane;
            v[u] = w < we ? nq_valid[w] : -1;
        }
#pragma unroll
        for (int u = 0; u < PO_V; ++u) {
            const int w = base + u * MSSVT_WAVE + lane;
            const int val = v[u] > 0 ? v[u] : 0;
            int incl = val;
            for (int off = 1; off < MSSVT_WAVE; off <<= 1) {
                const int t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            if (v[u] >= 0) {
                q_off[w] = carry + incl - val;
                const int k = min(v[u], max_key);
                if (k > 0) perm[key_base[k] + atomicAdd(&hist[wv][k], 1)] = w;
            }
            carry += __shfl(incl, MSSVT_WAVE - 1);
        }
    }
}

// compact query rows: row q_off[w] + p = the p-th valid slot of window w's query list:
// rmeta = the slot's (rel.xyz, bits(feature row)), rsrc = (window, attn row w * nq + slot)
__global__ void __launch_bounds__(256) k_query_rows(const int *num_wins, int row_capacity, PlanOrderPack pack) {
    const PlanOrderSet &ps = pack.s[blockIdx.y];
    const int nq = ps.nq;
    const float4 *qmeta = ps.qmeta;
    const int *q_off = ps.q_off;
    float4 *rmeta = ps.rmeta;
    int2 *rsrc = ps.rsrc;
    const int nw = *num_wins, lane = lane_id();
    const int wstep = gridDim.x * (blockDim.x / MSSVT_WAVE);
    for (int w = blockIdx.x * (blockDim.x / MSSVT_WAVE) + threadIdx.x / MSSVT_WAVE; w < nw; w += wstep) {
        int run = q_off[w];
        for (int q0 = 0; q0 < nq; q0 += MSSVT_WAVE) {
            const int qi = q0 + lane;
            float4 qm = make_float4(0.f, 0.f, 0.f, 0.f);
            bool ok = false;
            if (qi < nq) {
                qm = qmeta[(size_t)w * nq + qi];
                ok = __builtin_bit_cast(int, qm.w) >= 0;
            }
            const unsigned long long m = __ballot(ok);
            const int r = run + __popcll(m & ((1ull << lane) - 1ull));
            if (ok && r < row_capacity) {
                rmeta[r] = qm;
                rsrc[r] = make_int2(w, w * nq + qi);
            }
            run += __popcll(m);
        }
    }
}

extern "C" int mssvt_plan_order_multi(int num_sets, const int *num_wins_dev, const int *const *host_nq_valid,
                                      const int *host_nq, const float *const *host_qmeta, int win_capacity,
                                      int row_capacity, int *const *host_perm, int *const *host_num_active,
                                      int *const *host_q_off, float *const *host_qrow_meta,
                                      int *const *host_qrow_src, int *const *host_num_rows, void *stream) {
    if (num_sets <= 0 || num_sets > PO_MAX_SETS) return num_sets <= 0 ? MSSVT_E_BADARG : MSSVT_E_TOOLARGE;
    if (!num_wins_dev || !host_nq_valid || !host_nq || !host_qmeta || !host_perm || !host_num_active || !host_q_off ||
        !host_qrow_meta || !host_qrow_src || !host_num_rows || win_capacity <= 0 || row_capacity <= 0)
        return MSSVT_E_BADARG;
    PlanOrderPack pack;
    for (int i = 0; i < PO_MAX_SETS; ++i) {
        const int k = i < num_sets ? i : 0;
        if (!host_nq_valid[k] || !host_qmeta[k] || !host_perm[k] || !host_num_active[k] || !host_q_off[k] ||
            !host_qrow_meta[k] || !host_qrow_src[k] || !host_num_rows[k] || host_nq[k] <= 0)
            return MSSVT_E_BADARG;
        PlanOrderSet &ps = pack.s[i];
        ps.nq_valid = host_nq_valid[k];
        ps.nq = host_nq[k];
        ps.max_key = host_nq[k] > 256 ? 256 : host_nq[k];
        ps.qmeta = reinterpret_cast<const float4 *>(host_qmeta[k]);
        ps.perm = host_perm[k]; ps.num_active = host_num_active[k]; ps.q_off = host_q_off[k];
        ps.num_rows = host_num_rows[k];
        ps.rmeta = reinterpret_cast<float4 *>(host_qrow_meta[k]);
        ps.rsrc = reinterpret_cast<int2 *>(host_qrow_src[k]);
    }
    // workgroups per list: one per ~2k windows of capacity, as long as their histograms fit the head of the row array
    int G = win_capacity / 2048;
    if (G > 64) G = 64;
    while (G > 1 && (long long)G * (PO_KEYS + 1) > 2LL * row_capacity) --G;
    if (G > 1) {
        k_plan_order<0><<<dim3(G, num_sets), PO_WAVES * MSSVT_WAVE, 0, (hipStream_t)stream>>>(num_wins_dev, row_capacity, pack);
        k_plan_order<1><<<dim3(G, num_sets), PO_WAVES * MSSVT_WAVE, 0, (hipStream_t)stream>>>(num_wins_dev, row_capacity, pack);
    } else {
        k_plan_order<1><<<dim3(1, num_sets), PO_WAVES * MSSVT_WAVE, 0, (hipStream_t)stream>>>(num_wins_dev, row_capacity, pack);
    }
    int grid = (win_capacity + 3) / 4;
    if (grid > 4096) grid = 4096;
    k_query_rows<<<dim3(grid, num_sets), 256, 0, (hipStream_t)stream>>>(num_wins_dev, row_capacity, pack);
    return mssvt_launch_status();
}

extern "C" int mssvt_plan_order(const int *num_wins_dev, const int *nq_valid, int nq, const float *qmeta,
                                int win_capacity, int row_capacity, int *perm, int *num_active_dev, int *q_off,
                                float *qrow_meta, int *qrow_src, int *num_rows_dev, void *stream) {
    return mssvt_plan_order_multi(1, num_wins_dev, &nq_valid, &nq, &qmeta, win_capacity, row_capacity, &perm,
                                  &num_active_dev, &q_off, &qrow_meta, &qrow_src, &num_rows_dev, stream);
}
