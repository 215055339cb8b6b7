// block_attn.hip -- fused mixed-scale window attention of one MsSVT Block (fp32).
//
// Replaces, for one head group g of one Block, the reference's
//   7 x K5 feature/coordinate gathers           (ref: mssvt_backbone.py:260-268)
//   relative coordinates + positional MLP        (ref: :269-282, pos_proj :43-47)
//   MixedScaleAttention.forward for group g      (ref: mssvt_utils.py:112-150)
// and, in a second kernel, K9 + K10 + the interpolation weights + the per-sample
// index_put scatter + the first residual        (ref: mssvt_backbone.py:298-338).
// Nothing padded is written to HBM: per window only the valid query rows (x C/G
// channels) leave the kernel.
//
// Work decomposition.  Windows are tiny and ragged (160k-point scene: ~2 valid
// queries and ~4 + ~20 unmasked keys per window) while the projection weights are
// small (4 * Cg^2 floats per group), so the kernel is PERSISTENT: each workgroup
// stages the group's weights in LDS once (64 KiB at Cg = 64), then its wavefronts
// walk the windows, ONE WAVEFRONT PER WINDOW, lane = channel of the group's slice,
// so every feature row is read as one coalesced 256-B segment.
//
// Arithmetic is re-associated around the small side of the problem (#queries <<
// #keys): with q' = scale * (Wq x_q + b_q),
//   score_h(k)  = q'_h . (Wk_h x_k + bk_h) = (Wk_h^T q'_h) . x_k + const_h
//   out_h       = sum_k p_hk (Wv_h x_k + bv_h) = Wv_h (sum_k p_hk x_k) + bv_h
// (const_h cancels in the softmax, sum_k p_hk = 1), i.e. keys are never projected:
// per query 4 mat-vecs of size Cg^2, per (query,key) pair 2*heads*Cg MACs.  Masked
// key slots (additive -100 in the reference -> relative weight <= e^-100) are
// skipped; slot 0 of each scale is never masked, so no key set is empty.
// Differences to the reference are re-association only (~1e-6 relative).
//
// LDS (floats): WqT | Wk | WvT | WoT (Cg^2 each) | bq bv bo | per wave:
//   see k_block_attn below.
#include "common.hip.h"
#include <stdlib.h>

#define ATTN_MAX_WAVES 8

struct AttnArgs {
    int C, c0, heads, hd;
    float scale;
    int nq, K;
    const float *xhat;
    const int *num_wins;  // number of entries of `perm` (windows with at least one valid query)
    const int *perm;      // work order: heavy windows first
    // per-slot metadata resolved by the plan kernel (window_plan.hip): (rel.x, rel.y, rel.z,
    // bits(global feature row or -1)); wcentre = window centre in metres
    const float4 *qmeta, *kmeta, *wcentre;
    const float *Wq, *bq, *Wkv, *bkv, *Wo, *bo, *Wp, *bp;
    float *attn;
    int *work_counter;  // 128 device ints (8 counters, one per 64-B line), zeroed per launch
    int wave_floats;
    int dbg;
};

// cell centre in metres, one rounding per op like the reference's torch expression
// (ref: with_coords, mssvt_backbone.py:132-137)
__device__ __forceinline__ float centre_of(int idx, float cell, float lo) {
    return __fadd_rn(__fmul_rn(__fadd_rn((float)idx, 0.5f), cell), lo);
}


// y[lane] = bias + sum_i W4[i/4][lane][i%4] * x[i]: the matrix is stored so that one
// ds_read_b128 per lane brings 4 consecutive inputs' weights, x comes as a broadcast
// ds_read_b128; CG/4 fully unrolled steps keep ~2*CG/4 LDS reads in flight.
template <int CG>
__device__ __forceinline__ float matvec4(const float *W4, const float *x, int cl, float bias) {
    const float4 *w = reinterpret_cast<const float4 *>(W4) + cl;
    const float4 *xv = reinterpret_cast<const float4 *>(x);
    float a0 = bias, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    // 8 steps (16 ds_read_b128) are issued back to back, THEN consumed: with <= 2 waves per
    // SIMD nobody else hides the ~100-cycle LDS latency, and left alone hipcc keeps only
    // two reads in flight (s_waitcnt lgkmcnt(2) after every pair).
    constexpr int STEP = CG / 4 < 8 ? CG / 4 : 8;
#pragma unroll
    for (int b = 0; b < CG / 4; b += STEP) {
        float4 wv[STEP], xx[STEP];
#pragma unroll
        for (int i = 0; i < STEP; ++i) {
            wv[i] = w[(b + i) * CG];
            xx[i] = xv[b + i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < STEP; ++i) {
            a0 = __builtin_fmaf(wv[i].x, xx[i].x, a0);
            a1 = __builtin_fmaf(wv[i].y, xx[i].y, a1);
            a2 = __builtin_fmaf(wv[i].z, xx[i].z, a2);
            a3 = __builtin_fmaf(wv[i].w, xx[i].w, a3);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return (a0 + a1) + (a2 + a3);
}

// CG = channels of the head group (compile time, multiple of 4, <= 64), HP = padded
// head count (4 or 8).  LDS (floats): WqT4 | WkT4 | WvT4 | WoT4 (CG^2 each, the
// [i/4][lane][i%4] layout above) | bq bv bo | per wave: keys[K][CG+4], xq[CG], qp[CG],
// qt[HP][CG], pb[K][HP], xbar[HP][CG+4], vb[CG], krow[K].
template <int CG, int HD, int HP>
__global__ void __launch_bounds__(ATTN_MAX_WAVES *MSSVT_WAVE) k_block_attn(AttnArgs a) {
    extern __shared__ float4 lds4[];
    float *lds = reinterpret_cast<float *>(lds4);
    constexpr int CG2 = CG * CG, KS = CG + 4;
    float *WqT = lds, *WkT = WqT + CG2, *WvT = WkT + CG2, *WoT = WvT + CG2;
    float *bq = WoT + CG2, *bv = bq + CG, *bo = bv + CG;
    const int wv = threadIdx.x / MSSVT_WAVE, lane = lane_id();
    float *wbase = bo + CG + (size_t)wv * a.wave_floats;
    float *keys = wbase;
    float *xq = keys + a.K * KS;
    float *qp = xq + CG;
    float *qt = qp + CG;
    float *pb = qt + HP * CG;
    float *xbar = pb + a.K * HP;
    float *vb = xq;  // xq is dead once q' exists; v lives in the same CG floats
    float *krel = xbar + HP * KS;  // [K][3] key coordinates relative to the window centre
    float *qrel = krel + 3 * a.K;  // [nq][3]
    int *krow = reinterpret_cast<int *>(qrel + 3 * a.nq);
    int *qrow = krow + a.K;
    int *qslot = qrow + a.nq;

    // ---- stage this group's weights once per persistent workgroup ---------------------
    for (int e = threadIdx.x; e < CG2; e += blockDim.x) {
        const int o = e / CG, i = e % CG;  // nn.Linear weight [o][i]
        const int t_io = ((i >> 2) * CG + o) * 4 + (i & 3);  // input index i in the b128, lane = output o
        const int t_oi = ((o >> 2) * CG + i) * 4 + (o & 3);  // summed index o in the b128, lane = i
        WqT[t_io] = a.Wq[e];
        WkT[t_oi] = a.Wkv[e];        // rows [0,CG) of to_kvs = K projection; folded onto the query
        WvT[t_io] = a.Wkv[CG2 + e];  // rows [CG,2CG) = V projection
        WoT[t_io] = a.Wo[e];
    }
    for (int e = threadIdx.x; e < CG; e += blockDim.x) {
        bq[e] = a.bq[e];
        bv[e] = a.bkv[CG + e];
        bo[e] = a.bo[e];
    }
    __syncthreads();

    const bool act = lane < CG;
    const int cl = act ? lane : 0;
    float wp[6], bpv;  // positional MLP row of this lane's channel (ref pos_proj.0: (C,6,1))
#pragma unroll
    for (int t = 0; t < 6; ++t) wp[t] = a.Wp[(size_t)(a.c0 + cl) * 6 + t];
    bpv = a.bp[a.c0 + cl];
    const int nw = *a.num_wins;
    const bool two_heads = a.K <= 32;  // score pass: lane = key + 32 * (head & 1)
    const int my_h = cl / a.hd;
    const int heads = a.heads;

    // Windows are handed out dynamically: their cost varies by >10x (0..20 queries x 1..32
    // keys) and a static split leaves most waves idle at the end.  One counter would
    // serialise (~88 tickets/us chip-wide), so there are 8 counters on separate cache
    // lines; counter c owns the windows w == c (mod 8) and is drawn 2 windows at a time,
    // first by the workgroups with blockIdx == c (mod 8) (one XCD under round-robin
    // placement; speed only), then by anybody (work stealing).  The ticket for the NEXT
    // pair is requested while the current pair is being processed, and the next window's
    // metadata is loaded while the current window computes, so neither latency is exposed.
    constexpr int TPA = 1;  // windows per ticket: 1 keeps the tail short (a wave holds <= 2 windows)
    int shard = blockIdx.x & 7, tries = 0, pair_lo = 0, pair_pos = TPA, pend = 0;
    bool have_pend = false;
    auto ticket_async = [&]() {
        int t = 0;
        if (lane == 0) t = atomicAdd(a.work_counter + 16 * shard, TPA);
        return t;
    };
    auto next_window = [&]() -> int {
        for (;;) {
            if (pair_pos >= TPA) {
                if (!have_pend) pend = ticket_async();
                pair_lo = __builtin_amdgcn_readfirstlane(pend);  // waits for the atomic issued a pair ago
                pend = ticket_async();
                have_pend = true;
                pair_pos = 0;
            }
            const int t = (pair_lo + pair_pos) * 8 + shard;
            ++pair_pos;
            if (t < nw) return a.perm[t];
            if (++tries >= 8) return -1;  // every shard drained
            shard = (shard + 1) & 7;
            pair_pos = TPA;
            have_pend = false;
        }
    };
    const float4 none4 = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, -1));
    // metadata of a window, one slot per lane (first 64 query slots; the rest is read on demand)
    auto load_meta = [&](int w, float4 &km, float4 &qm, float4 &wc) {
        km = lane < a.K ? a.kmeta[(size_t)w * a.K + lane] : none4;
        qm = lane < a.nq ? a.qmeta[(size_t)w * a.nq + lane] : none4;
        wc = a.wcentre[w];
    };

    int w = next_window();
    float4 km_c = none4, qm_c = none4, wc_c = none4;
    if (w >= 0) load_meta(w, km_c, qm_c, wc_c);
    while (w >= 0) {
        int w_next = -1;
        float4 km_n = none4, qm_n = none4, wc_n = none4;
        if (!(a.dbg & 1)) {
            w_next = next_window();
            if (w_next >= 0) load_meta(w_next, km_n, qm_n, wc_n);
        }

        const float cxm = wc_c.x, cym = wc_c.y, czm = wc_c.z;
        const float posc = bpv + wp[3] * cxm + wp[4] * cym + wp[5] * czm;  // window part of the pos. MLP
        // ---- unmasked keys / valid queries -> compact lists {row, rel. coordinates} ---------------
        int nkv = 0, nqv = 0;
        {
            const int row = __builtin_bit_cast(int, km_c.w);
            const bool ok = row >= 0;
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const int p = __popcll(m & ((1ull << lane) - 1ull));
                krow[p] = row;
                krel[3 * p + 0] = km_c.x;
                krel[3 * p + 1] = km_c.y;
                krel[3 * p + 2] = km_c.z;
            }
            nkv = __popcll(m);
        }
        for (int q0 = 0; q0 < a.nq; q0 += MSSVT_WAVE) {
            const int qi = q0 + lane;
            float4 qm = qm_c;
            if (q0 > 0) qm = qi < a.nq ? a.qmeta[(size_t)w * a.nq + qi] : none4;
            const int row = __builtin_bit_cast(int, qm.w);
            const bool ok = row >= 0;
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const int p = nqv + __popcll(m & ((1ull << lane) - 1ull));
                qrow[p] = row;
                qslot[p] = qi;
                qrel[3 * p + 0] = qm.x;
                qrel[3 * p + 1] = qm.y;
                qrel[3 * p + 2] = qm.z;
            }
            nqv += __popcll(m);
        }
        wave_lds_sync();
        if (nqv > 0) {
        // ---- key tokens: LN'd feature slice + positional embedding -> LDS; 16 row loads in
        //      flight per step (one latency per 16 keys) --------------------------------------
        for (int jb = 0; jb < nkv; jb += 16) {
            // unconditional loads (index clamped to the last key): a guarded load makes hipcc
            // branch around it and wait for each one in turn
            float val[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int jc = min(jb + u, nkv - 1);
                val[u] = a.xhat[(size_t)krow[jc] * a.C + a.c0 + cl];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int jj = jb + u, jc = min(jj, nkv - 1);
                const float pos = fmaxf(posc + wp[0] * krel[3 * jc] + wp[1] * krel[3 * jc + 1] +
                                        wp[2] * krel[3 * jc + 2], 0.0f);
                if (jj < nkv && act) keys[jj * KS + lane] = val[u] + pos;
            }
        }
        // ---- queries (their rows are fetched 4 at a time) ---------------------------------------
        for (int qb = 0; qb < nqv; qb += 4) {
          float qval[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
              qval[u] = a.xhat[(size_t)qrow[min(qb + u, nqv - 1)] * a.C + a.c0 + cl];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int qq = qb + u;
            if (qq >= nqv) break;
            const int qi = qslot[qq];
            const float pos = fmaxf(posc + wp[0] * qrel[3 * qq] + wp[1] * qrel[3 * qq + 1] + wp[2] * qrel[3 * qq + 2], 0.0f);
            if (act) xq[lane] = qval[u] + pos;
            wave_lds_sync();
            // q' = Wq xq + bq                       (lane = output channel)
            // q' = Wq xq + bq                       (lane = output channel)
            const float qpv = matvec4<CG>(WqT, xq, cl, bq[cl]);
            if (act) qp[lane] = qpv;
            wave_lds_sync();
            // qt_h = scale * Wk_h^T q'_h            (lane = input channel; all heads, fully unrolled)
            {
                const float4 *wk = reinterpret_cast<const float4 *>(WkT) + cl;
                const float4 *qv = reinterpret_cast<const float4 *>(qp);
                float acc[CG / HD];
#pragma unroll
                for (int h = 0; h < CG / HD; ++h) acc[h] = 0.f;
                constexpr int STEP = CG / 4 < 8 ? CG / 4 : 8;
#pragma unroll
                for (int b = 0; b < CG / 4; b += STEP) {
                    float4 wv4[STEP], q4[STEP];
#pragma unroll
                    for (int i = 0; i < STEP; ++i) {
                        wv4[i] = wk[(b + i) * CG];
                        q4[i] = qv[b + i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < STEP; ++i) {
                        const int h = ((b + i) * 4) / HD;  // compile-time after unrolling
                        acc[h] = __builtin_fmaf(wv4[i].x, q4[i].x, acc[h]);
                        acc[h] = __builtin_fmaf(wv4[i].y, q4[i].y, acc[h]);
                        acc[h] = __builtin_fmaf(wv4[i].z, q4[i].z, acc[h]);
                        acc[h] = __builtin_fmaf(wv4[i].w, q4[i].w, acc[h]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (act) {
#pragma unroll
                    for (int h = 0; h < CG / HD; ++h) qt[h * CG + lane] = acc[h] * a.scale;
                }
            }
            wave_lds_sync();
            // scores + softmax                      (lane = key, two heads side by side when K <= 32)
            {
                const int j = two_heads ? (lane & 31) : lane;
                const int npass = two_heads ? (heads + 1) / 2 : heads;
                const float4 *kr = reinterpret_cast<const float4 *>(keys + (j < nkv ? j : 0) * KS);
                for (int p = 0; p < npass; ++p) {
                    const int h = two_heads ? 2 * p + (lane >> 5) : p;
                    const bool on = j < nkv && h < heads;
                    const float4 *qh = reinterpret_cast<const float4 *>(qt + (h < heads ? h : 0) * CG);
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                    constexpr int STEP = CG / 4 < 8 ? CG / 4 : 8;
#pragma unroll
                    for (int b = 0; b < CG / 4; b += STEP) {
                        float4 kk[STEP], qq[STEP];
#pragma unroll
                        for (int i = 0; i < STEP; ++i) {
                            kk[i] = kr[b + i];
                            qq[i] = qh[b + i];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < STEP; ++i) {
                            s0 = __builtin_fmaf(kk[i].x, qq[i].x, s0);
                            s1 = __builtin_fmaf(kk[i].y, qq[i].y, s1);
                            s2 = __builtin_fmaf(kk[i].z, qq[i].z, s2);
                            s3 = __builtin_fmaf(kk[i].w, qq[i].w, s3);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const float sc = on ? (s0 + s1) + (s2 + s3) : -INFINITY;
                    const float mx = two_heads ? half_max(sc) : wave_max(sc);
                    const float e = on ? __expf(sc - mx) : 0.0f;
                    const float sum = two_heads ? half_sum(e) : wave_sum(e);
                    if (j < a.K && h < heads) pb[j * HP + h] = e * __builtin_amdgcn_rcpf(sum);  // 0 for unused rows
                }
            }
            wave_lds_sync();
            // xbar_h = sum_k p_hk x_k                (lane = channel, all heads at once, 8 keys per step)
            {
                float acc[HP];
#pragma unroll
                for (int h = 0; h < HP; ++h) acc[h] = 0.f;
                for (int jb = 0; jb < nkv; jb += 8) {
                    float kv[8];
                    float4 pp[8][HP / 4];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int jj = jb + u, jc = jj < nkv ? jj : nkv - 1;  // rows >= nkv: weight 0 (see above)
                        kv[u] = keys[jc * KS + cl];
                        const int jp = jj < a.K ? jj : a.K - 1;
#pragma unroll
                        for (int h4 = 0; h4 < HP / 4; ++h4) pp[u][h4] = reinterpret_cast<const float4 *>(pb + jp * HP)[h4];
                        if (jj >= a.K) {
#pragma unroll
                            for (int h4 = 0; h4 < HP / 4; ++h4) pp[u][h4] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
#pragma unroll
                        for (int h4 = 0; h4 < HP / 4; ++h4) {
                            acc[4 * h4 + 0] = __builtin_fmaf(pp[u][h4].x, kv[u], acc[4 * h4 + 0]);
                            acc[4 * h4 + 1] = __builtin_fmaf(pp[u][h4].y, kv[u], acc[4 * h4 + 1]);
                            acc[4 * h4 + 2] = __builtin_fmaf(pp[u][h4].z, kv[u], acc[4 * h4 + 2]);
                            acc[4 * h4 + 3] = __builtin_fmaf(pp[u][h4].w, kv[u], acc[4 * h4 + 3]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (act) {
#pragma unroll
                    for (int h = 0; h < HP; ++h)
                        if (h < heads) xbar[h * KS + lane] = acc[h];
                }
            }
            wave_lds_sync();
            // v = Wv xbar_{head(o)} + bv             (lane = output channel o)
            const float vbv = matvec4<CG>(WvT, xbar + my_h * KS, cl, bv[cl]);
            if (act) vb[lane] = vbv;
            wave_lds_sync();
            // out = Wo v + bo
            const float out = matvec4<CG>(WoT, vb, cl, bo[cl]);
            if (act) a.attn[((size_t)w * a.nq + qi) * a.C + a.c0 + lane] = out;
            wave_lds_sync();  // xq (= vb) is rewritten by the next query
          }
        }
        }  // nqv > 0
        wave_lds_sync();  // keys / lists are rewritten for the next window
        if (a.dbg & 1) {
            w_next = next_window();
            if (w_next >= 0) load_meta(w_next, km_n, qm_n, wc_n);
        }
        w = w_next;
        km_c = km_n;
        qm_c = qm_n;
        wc_c = wc_n;
    }
}

template <int CG, int HD, int HP>
static int launch_block_attn(AttnArgs &a, hipStream_t stream) {
    constexpr int KS = CG + 4;
    a.wave_floats = a.K * KS + 2 * CG + HP * CG + a.K * HP + HP * KS + 4 * a.K + 5 * a.nq;
    a.wave_floats = (a.wave_floats + 3) & ~3;  // keep every wave's region 16-B aligned
    const size_t fixed = (size_t)4 * CG * CG + 3 * CG;
    int waves = ATTN_MAX_WAVES;
    while (waves > 1 && (fixed + (size_t)waves * a.wave_floats) * 4 > 160 * 1024) --waves;
    const size_t lds_bytes = (fixed + (size_t)waves * a.wave_floats) * 4;
    if (lds_bytes > 160 * 1024) return MSSVT_E_TOOLARGE;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_block_attn<CG, HD, HP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    // persistent grid: as many workgroups per CU as the LDS footprint admits (256 CUs on MI355X)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    int per_cu = (int)((160 * 1024) / lds_bytes);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    hipError_t me = hipMemsetAsync(a.work_counter, 0, 128 * sizeof(int), stream);
    if (me != hipSuccess) return (int)me;
    k_block_attn<CG, HD, HP><<<cus * per_cu, waves * MSSVT_WAVE, lds_bytes, stream>>>(a);
    return mssvt_launch_status();
}

extern "C" int mssvt_block_attention_group(
    int C, int c0, int Cg, int heads, int head_dim, float scale, int nq, int key_num_sample,
    const float *xhat, const int *num_active_dev, const int *perm, const float *qmeta, const float *kmeta,
    const float *wcentre, const float *Wq, const float *bq, const float *Wkv, const float *bkv, const float *Wo, const float *bo,
    const float *Wpos, const float *bpos, float *attn, int *work_counter, void *stream) {
    if (!work_counter) return MSSVT_E_BADARG;
    if (!xhat || !num_active_dev || !perm || !qmeta || !kmeta || !wcentre || !Wq || !bq || !Wkv || !bkv || !Wo || !bo ||
        !Wpos || !bpos || !attn || C <= 0 || Cg <= 0 || heads <= 0 || head_dim <= 0 || nq <= 0 ||
        key_num_sample <= 0)
        return MSSVT_E_BADARG;
    if (Cg != heads * head_dim || c0 < 0 || c0 + Cg > C) return MSSVT_E_BADARG;
    // one channel per lane, heads aligned to 4-float LDS vectors, <= 8 heads per group
    if (Cg > MSSVT_WAVE || key_num_sample > MSSVT_WAVE || (head_dim & 3) || heads > 8) return MSSVT_E_TOOLARGE;
    AttnArgs a;
    a.C = C; a.c0 = c0; a.heads = heads; a.hd = head_dim; a.scale = scale;
    a.nq = nq; a.K = key_num_sample;
    a.xhat = xhat; a.num_wins = num_active_dev; a.perm = perm;
    a.qmeta = reinterpret_cast<const float4 *>(qmeta);
    a.kmeta = reinterpret_cast<const float4 *>(kmeta);
    a.wcentre = reinterpret_cast<const float4 *>(wcentre);
    a.Wq = Wq; a.bq = bq; a.Wkv = Wkv; a.bkv = bkv; a.Wo = Wo; a.bo = bo; a.Wp = Wpos; a.bp = bpos;
    a.attn = attn;
    a.work_counter = work_counter;
    a.dbg = getenv("MSSVT_DBG") ? atoi(getenv("MSSVT_DBG")) : 0;
    hipStream_t st = (hipStream_t)stream;
#define MSSVT_ATTN_CASE(cg, hd)                                   \
    if (Cg == cg && head_dim == hd)                               \
        return launch_block_attn<cg, hd, ((cg / hd + 3) / 4) * 4>(a, st);
    MSSVT_ATTN_CASE(8, 8)
    MSSVT_ATTN_CASE(16, 8)
    MSSVT_ATTN_CASE(16, 16)
    MSSVT_ATTN_CASE(24, 8)
    MSSVT_ATTN_CASE(32, 8)
    MSSVT_ATTN_CASE(32, 16)
    MSSVT_ATTN_CASE(32, 32)
    MSSVT_ATTN_CASE(48, 16)
    MSSVT_ATTN_CASE(64, 8)
    MSSVT_ATTN_CASE(64, 16)
    MSSVT_ATTN_CASE(64, 32)
    return MSSVT_E_TOOLARGE;  // shape not instantiated: the caller falls back to the operator path
#undef MSSVT_ATTN_CASE
}

// ---------------------------------------------------------------------------------
// interpolation (3-NN, inverse distance) + scatter + first residual
// ---------------------------------------------------------------------------------
struct ScatterArgs {
    int C, nq, n1, interp;
    const float *attn, *x_in;
    float *x_new;
    const int *indices, *win_ind, *num_wins, *win_vstart, *q_ind, *upd_ind, *owner;
    float vsx, vsy, vsz, minx, miny, minz;
    // table mode (tab_row != null): nothing is gathered; per owned voxel the three attention
    // rows and weights are recorded so that a consumer (the fused FFN) can apply them
    int4 *tab_row;
    float4 *tab_w;
    int zero_row;  // row of `attn` that holds zeros: target of empty slots / zero weights
};

#define SC_WPB 4
#define SC_MAXQ 256

__global__ void __launch_bounds__(SC_WPB *MSSVT_WAVE) k_block_scatter(ScatterArgs a) {
    __shared__ float kx[SC_WPB][SC_MAXQ], ky[SC_WPB][SC_MAXQ], kz[SC_WPB][SC_MAXQ];
    __shared__ int kvalid[SC_WPB][SC_MAXQ];
    const int wv = threadIdx.x / MSSVT_WAVE, lane = lane_id();
    const int nw = *a.num_wins;
    for (int w = blockIdx.x * SC_WPB + wv; w < nw; w += gridDim.x * SC_WPB) {
        const int vstart = a.win_vstart[w];
        if (!a.interp) {  // ref mssvt_backbone.py:327-330: only the query voxels are updated
            for (int i = 0; i < a.nq; ++i) {
                const int v = a.q_ind[(size_t)w * a.nq + i];
                if (v < 0 || a.owner[vstart + v] != w * a.nq + i) continue;
                if (a.tab_row) {
                    if (lane == 0) {
                        a.tab_row[vstart + v] = make_int4(w * a.nq + i, a.zero_row, a.zero_row, 0);
                        a.tab_w[vstart + v] = make_float4(1.f, 0.f, 0.f, 0.f);
                    }
                    continue;
                }
                const float *src = a.attn + ((size_t)w * a.nq + i) * a.C;
                const size_t row = (size_t)(vstart + v) * a.C;
                for (int c = lane; c < a.C; c += MSSVT_WAVE) a.x_new[row + c] = src[c] + a.x_in[row + c];
            }
            continue;
        }
        // known points = ALL nq query slots; empty slots sit at the world origin with zero
        // features (ref :302 gathers coordinates with -1 -> 0 fill) -- kept as is
        for (int i = lane; i < a.nq; i += MSSVT_WAVE) {
            const int v = a.q_ind[(size_t)w * a.nq + i];
            float x = 0.f, y = 0.f, z = 0.f;
            if (v >= 0) {
                const int4 vi = reinterpret_cast<const int4 *>(a.indices)[vstart + v];
                x = centre_of(vi.w, a.vsx, a.minx);
                y = centre_of(vi.z, a.vsy, a.miny);
                z = centre_of(vi.y, a.vsz, a.minz);
            }
            kx[wv][i] = x; ky[wv][i] = y; kz[wv][i] = z;
            kvalid[wv][i] = v >= 0;
        }
        wave_lds_sync();
        for (int s0 = 0; s0 < a.n1; s0 += MSSVT_WAVE) {
            const int s = s0 + lane;
            int v = -1;
            if (s < a.n1) {
                v = a.upd_ind[(size_t)w * a.n1 + s];
                if (v >= 0 && a.owner[vstart + v] != w * a.n1 + s) v = -1;  // another slot owns this voxel
            }
            int i1 = 0, i2 = 0, i3 = 0;
            float w1 = 0.f, w2 = 0.f, w3 = 0.f;
            if (v >= 0) {  // K9 (ref interpolate_gpu.cu:16-59) + weights (ref mssvt_backbone.py:305-307)
                const int4 vi = reinterpret_cast<const int4 *>(a.indices)[vstart + v];
                const float ux = centre_of(vi.w, a.vsx, a.minx), uy = centre_of(vi.z, a.vsy, a.miny),
                            uz = centre_of(vi.y, a.vsz, a.minz);
                double b1 = 1e40, b2 = 1e40, b3 = 1e40;
                for (int k = 0; k < a.nq; ++k) {
                    const float dx = ux - kx[wv][k], dy = uy - ky[wv][k], dz = uz - kz[wv][k];
                    const float d = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
                    else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
                    else if (d < b3) { b3 = d; i3 = k; }
                }
                const float d1 = fmaxf(sqrtf((float)b1), 1e-10f), d2 = fmaxf(sqrtf((float)b2), 1e-10f),
                            d3 = fmaxf(sqrtf((float)b3), 1e-10f);
                w1 = 1.0f / d1; w2 = 1.0f / d2; w3 = 1.0f / d3;
                const float norm = (w1 + w2) + w3;
                w1 /= norm; w2 /= norm; w3 /= norm;
                if (!kvalid[wv][i1]) w1 = 0.f;  // empty slots carry zero features
                if (!kvalid[wv][i2]) w2 = 0.f;
                if (!kvalid[wv][i3]) w3 = 0.f;
            }
            if (a.tab_row) {
                if (v >= 0) {
                    a.tab_row[vstart + v] = make_int4(w1 != 0.f ? w * a.nq + i1 : a.zero_row,
                                                      w2 != 0.f ? w * a.nq + i2 : a.zero_row,
                                                      w3 != 0.f ? w * a.nq + i3 : a.zero_row, 0);
                    a.tab_w[vstart + v] = make_float4(w1, w2, w3, 0.f);
                }
                continue;
            }
            unsigned long long todo = __ballot(v >= 0);
            while (todo) {  // one covered voxel at a time, lanes sweep its channels
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int vv = __shfl(v, src);
                const int j1 = __shfl(i1, src), j2 = __shfl(i2, src), j3 = __shfl(i3, src);
                const float f1 = __shfl(w1, src), f2 = __shfl(w2, src), f3 = __shfl(w3, src);
                const float *r1 = a.attn + ((size_t)w * a.nq + j1) * a.C;
                const float *r2 = a.attn + ((size_t)w * a.nq + j2) * a.C;
                const float *r3 = a.attn + ((size_t)w * a.nq + j3) * a.C;
                const size_t row = (size_t)(vstart + vv) * a.C;
                for (int c = lane; c < a.C; c += MSSVT_WAVE) {
                    float acc = 0.f;  // a zero weight never touches the (unwritten) row of an empty slot
                    if (f1 != 0.f) acc = r1[c] * f1;
                    if (f2 != 0.f) acc += r2[c] * f2;
                    if (f3 != 0.f) acc += r3[c] * f3;
                    a.x_new[row + c] = acc + a.x_in[row + c];
                }
            }
        }
        wave_lds_sync();
    }
}

extern "C" int mssvt_block_interp_scatter(int C, int nq, int n_upd, int use_interpolation,
                                          const float *attn, const float *x_in, float *x_new,
                                          const int *indices, const int *win_ind,
                                          const int *num_wins_dev, int win_capacity,
                                          const int *win_vstart, const int *q_ind,
                                          const int *upd_ind, const int *owner,
                                          const float *host_voxel_size3,
                                          const float *host_range_min3, void *stream) {
    if (!attn || !x_in || !x_new || !indices || !win_ind || !num_wins_dev || !win_vstart || !q_ind ||
        !owner || !host_voxel_size3 || !host_range_min3 || C <= 0 || nq <= 0)
        return MSSVT_E_BADARG;
    if (use_interpolation && (!upd_ind || n_upd <= 0)) return MSSVT_E_BADARG;
    if (nq > SC_MAXQ) return MSSVT_E_TOOLARGE;
    if (win_capacity <= 0) return MSSVT_OK;
    ScatterArgs a;
    a.C = C; a.nq = nq; a.n1 = n_upd; a.interp = use_interpolation;
    a.attn = attn; a.x_in = x_in; a.x_new = x_new;
    a.indices = indices; a.win_ind = win_ind; a.num_wins = num_wins_dev; a.win_vstart = win_vstart;
    a.q_ind = q_ind; a.upd_ind = upd_ind; a.owner = owner;
    a.vsx = host_voxel_size3[0]; a.vsy = host_voxel_size3[1]; a.vsz = host_voxel_size3[2];
    a.minx = host_range_min3[0]; a.miny = host_range_min3[1]; a.minz = host_range_min3[2];
    a.tab_row = nullptr; a.tab_w = nullptr; a.zero_row = 0;
    int grid = divup(win_capacity, SC_WPB);
    if (grid > 4096) grid = 4096;  // grid-stride over the windows actually present
    k_block_scatter<<<grid, SC_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(a);
    return mssvt_launch_status();
}

// Table form of mssvt_block_interp_scatter: records, per voxel owned by a list slot, the
// (up to) three attention rows and inverse-distance weights instead of applying them.
extern "C" int mssvt_block_interp_table(int nq, int n_upd, int use_interpolation, const int *indices,
                                        const int *win_ind, const int *num_wins_dev, int win_capacity,
                                        const int *win_vstart, const int *q_ind, const int *upd_ind,
                                        const int *owner, const float *host_voxel_size3,
                                        const float *host_range_min3, int zero_row, int *tab_row,
                                        float *tab_w, void *stream) {
    if (!indices || !win_ind || !num_wins_dev || !win_vstart || !q_ind || !owner || !host_voxel_size3 ||
        !host_range_min3 || !tab_row || !tab_w || nq <= 0)
        return MSSVT_E_BADARG;
    if (use_interpolation && (!upd_ind || n_upd <= 0)) return MSSVT_E_BADARG;
    if (nq > SC_MAXQ) return MSSVT_E_TOOLARGE;
    if (win_capacity <= 0) return MSSVT_OK;
    ScatterArgs a;
    a.C = 0; a.nq = nq; a.n1 = n_upd; a.interp = use_interpolation;
    a.attn = nullptr; a.x_in = nullptr; a.x_new = nullptr;
    a.indices = indices; a.win_ind = win_ind; a.num_wins = num_wins_dev; a.win_vstart = win_vstart;
    a.q_ind = q_ind; a.upd_ind = upd_ind; a.owner = owner;
    a.vsx = host_voxel_size3[0]; a.vsy = host_voxel_size3[1]; a.vsz = host_voxel_size3[2];
    a.minx = host_range_min3[0]; a.miny = host_range_min3[1]; a.minz = host_range_min3[2];
    a.tab_row = reinterpret_cast<int4 *>(tab_row);
    a.tab_w = reinterpret_cast<float4 *>(tab_w);
    a.zero_row = zero_row;
    int grid = divup(win_capacity, SC_WPB);
    if (grid > 4096) grid = 4096;
    k_block_scatter<<<grid, SC_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(a);
    return mssvt_launch_status();
}
