// block_attn.hip -- fused mixed-scale window attention of one MsSVT Block (fp32).
//
// Replaces, for one head group g of one Block, the reference's
//   7 x K5 feature/coordinate gathers           (ref: mssvt_backbone.py:260-268)
//   relative coordinates + positional MLP        (ref: :269-282, pos_proj :43-47)
//   MixedScaleAttention.forward for group g      (ref: mssvt_utils.py:112-150)
// and, in the table kernel at the end of this file, K9 + K10 + the interpolation weights
// (ref: mssvt_backbone.py:298-311).  Nothing padded is written to HBM.
//
// Arithmetic is re-associated around the small side of the problem (#queries << #keys: ~2 valid
// queries against ~4 + ~20 unmasked keys per window at 160k points): with q' = Wq x_q + b_q,
//   score_h(k)  = scale q'_h . (Wk_h x_k + bk_h) = (scale Wk_h^T q'_h) . x_k + const_h
//   out_h       = sum_k p_hk (Wv_h x_k + bv_h)   = Wv_h (sum_k p_hk x_k) + bv_h
// (const_h cancels in the softmax, sum_k p_hk = 1): keys are never projected; per query 4 mat-vecs
// of size Cg^2, per (query,key) pair 2*heads*Cg MACs.  Masked key slots (additive -100 in the
// reference -> relative weight <= e^-100) are skipped; slot 0 of each scale is never masked, so
// no key set is empty.  Differences to the reference are re-association only (~1e-6 relative).
//
// THREE PHASES, one launch each, all "one wavefront per window, lane = channel":
//   A  queries : x_q = xhat row + pos. embedding; q' = Wq x_q + b; qt_h = scale Wk_h^T q'_h -> qbuf
//   B  keys    : key tokens (xhat rows + pos. embedding) -> LDS; per query scores, softmax,
//                xbar_h = sum_k p_hk x_k -> qbuf (in place of qt)
//   C  output  : v = Wv xbar_{head} + bv; out = Wo v + bo -> attn rows
// A single fused kernel (first version) needs all four Cg x Cg matrices (64 KiB) PLUS the key tile
// (9 KiB per wave) in LDS: 7-8 waves per CU, every LDS / DPP / gather latency exposed (all pipes
// ~25 % busy, 190 us per launch).  Split, A and C hold two matrices and ~1 KiB per wave, B holds no
// weights at all: 16+ waves per CU each.  The price is one 4*HP*Cg-byte row per query and group
// written by A, rewritten by B, read by C (qbuf, L2/MALL resident).
//
// Windows are processed in the plan's work order (heaviest first, mssvt_plan_order), dealt
// round-robin to the wavefronts of a persistent grid.
#include "common.hip.h"

#define ATTN_MAX_WAVES 16

struct AttnArgs {
    int C, c0, heads, hd;
    float scale;
    int nq, K;
    const float *xhat;
    const int *num_wins;  // number of entries of `perm` (windows with at least one valid query)
    const int *perm;      // work order: heavy windows first
    const int *q_off;     // (cap) first compact query row of each window
    // per-slot metadata resolved by the plan kernel (window_plan.hip): (rel.x, rel.y, rel.z,
    // bits(global feature row or -1)); wcentre = window centre in metres
    const float4 *qmeta, *kmeta, *wcentre;
    const float *Wq, *bq, *Wkv, *bkv, *Wo, *bo, *Wp, *bp;
    float *qbuf;  // (query rows, HP*CG): qt after phase A, xbar after phase B
    float *attn;
    int wave_floats;
};

// y[lane] = bias + sum_i W4[i/4][lane][i%4] * x[i]: the matrix is stored so that one
// ds_read_b128 per lane brings 4 consecutive inputs' weights, x comes as a broadcast
// ds_read_b128; 8 steps (16 reads) are issued back to back, THEN consumed (left alone hipcc
// keeps only two reads in flight: s_waitcnt lgkmcnt(2) after every pair).
template <int CG>
__device__ __forceinline__ float matvec4(const float *W4, const float *x, int cl, float bias) {
    const float4 *w = reinterpret_cast<const float4 *>(W4) + cl;
    const float4 *xv = reinterpret_cast<const float4 *>(x);
    float a0 = bias, a1 = 0.f, a2 = 0.f, a3 = 0.f;
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

// PHASE 0 = A (queries), 1 = B (keys / softmax), 2 = C (output projections)
template <int CG, int HD, int HP, int PHASE>
__global__ void __launch_bounds__(ATTN_MAX_WAVES *MSSVT_WAVE) k_attn_phase(AttnArgs a) {
    extern __shared__ float4 lds4[];
    float *lds = reinterpret_cast<float *>(lds4);
    constexpr int CG2 = CG * CG, KS = CG + 4, NH = CG / HD;
    constexpr int QROW = HP * CG;  // floats per (query, group) row of qbuf
    const int wv = threadIdx.x / MSSVT_WAVE, lane = lane_id();
    const bool act = lane < CG;
    const int cl = act ? lane : 0;
    // ---- weights of this phase, once per persistent workgroup ------------------------------
    float *W0 = lds, *W1 = W0 + CG2, *b0 = W1 + CG2, *b1 = b0 + CG;
    float *wbase = (PHASE == 1 ? lds : b1 + CG) + (size_t)wv * a.wave_floats;
    if (PHASE != 1) {
        for (int e = threadIdx.x; e < CG2; e += blockDim.x) {
            const int o = e / CG, i = e % CG;  // nn.Linear weight [o][i]
            const int t_io = ((i >> 2) * CG + o) * 4 + (i & 3);  // input index i in the b128, lane = output o
            const int t_oi = ((o >> 2) * CG + i) * 4 + (o & 3);  // summed index o in the b128, lane = i
            if (PHASE == 0) {
                W0[t_io] = a.Wq[e];   // q' = Wq x
                W1[t_oi] = a.Wkv[e];  // rows [0,CG) of to_kvs = K projection, folded onto the query
            } else {
                W0[t_io] = a.Wkv[CG2 + e];  // rows [CG,2CG) = V projection
                W1[t_io] = a.Wo[e];
            }
        }
        for (int e = threadIdx.x; e < CG; e += blockDim.x) {
            b0[e] = PHASE == 0 ? a.bq[e] : a.bkv[CG + e];
            b1[e] = PHASE == 0 ? 0.f : a.bo[e];
        }
        __syncthreads();
    }
    // static round-robin over the heaviest-first work order: wave i takes entries i, i + T, i + 2T ...
    // (T = waves in the grid), i.e. one window of every weight tier -- as balanced as dynamic tickets
    // without their atomics (a drained single-address ticket costs ~11 ns chip-wide, x 8192 waves)
    const int n_act = *a.num_wins;
    const int wstep = gridDim.x * (blockDim.x / MSSVT_WAVE);
    const int wfirst = blockIdx.x * (blockDim.x / MSSVT_WAVE) + wv;
    const float4 none4 = make_float4(0.f, 0.f, 0.f, __builtin_bit_cast(float, -1));

    if (PHASE == 0) {
        // =========================== A: queries -> qt ==========================================
        float *xq = wbase, *qp = xq + CG;
        float *qrel = qp + CG;
        int *qrow = reinterpret_cast<int *>(qrel + 3 * a.nq);
        float wp[6], bpv;  // positional MLP row of this lane's channel (ref pos_proj.0: (C,6,1))
#pragma unroll
        for (int t = 0; t < 6; ++t) wp[t] = a.Wp[(size_t)(a.c0 + cl) * 6 + t];
        bpv = a.bp[a.c0 + cl];
        for (int wi = wfirst; wi < n_act; wi += wstep) {
            const int w = a.perm[wi];
            const float4 wc = a.wcentre[w];
            const float posc = bpv + wp[3] * wc.x + wp[4] * wc.y + wp[5] * wc.z;  // window part of the pos. MLP
            int nqv = 0;
            for (int q0 = 0; q0 < a.nq; q0 += MSSVT_WAVE) {
                const int qi = q0 + lane;
                const float4 qm = qi < a.nq ? a.qmeta[(size_t)w * a.nq + qi] : none4;
                const int row = __builtin_bit_cast(int, qm.w);
                const bool ok = row >= 0;
                const unsigned long long m = __ballot(ok);
                if (ok) {
                    const int p = nqv + __popcll(m & ((1ull << lane) - 1ull));
                    qrow[p] = row;
                    qrel[3 * p + 0] = qm.x;
                    qrel[3 * p + 1] = qm.y;
                    qrel[3 * p + 2] = qm.z;
                }
                nqv += __popcll(m);
            }
            wave_lds_sync();
            const size_t qbase = (size_t)a.q_off[w];
            for (int qb = 0; qb < nqv; qb += 4) {
                float qval[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)  // unconditional (clamped) loads: no branch + wait per element
                    qval[u] = a.xhat[(size_t)qrow[min(qb + u, nqv - 1)] * a.C + a.c0 + cl];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int qq = qb + u;
                    if (qq >= nqv) break;
                    const float pos = fmaxf(posc + wp[0] * qrel[3 * qq] + wp[1] * qrel[3 * qq + 1] + wp[2] * qrel[3 * qq + 2], 0.0f);
                    if (act) xq[lane] = qval[u] + pos;
                    wave_lds_sync();
                    const float qpv = matvec4<CG>(W0, xq, cl, b0[cl]);  // q' = Wq xq + bq (lane = output)
                    if (act) qp[lane] = qpv;
                    wave_lds_sync();
                    // qt_h = scale * Wk_h^T q'_h   (lane = input channel; all heads, fully unrolled)
                    const float4 *wk = reinterpret_cast<const float4 *>(W1) + cl;
                    const float4 *qv = reinterpret_cast<const float4 *>(qp);
                    float acc[NH];
#pragma unroll
                    for (int h = 0; h < NH; ++h) acc[h] = 0.f;
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
                        float *dst = a.qbuf + (qbase + qq) * QROW + lane;
#pragma unroll
                        for (int h = 0; h < NH; ++h) dst[h * CG] = acc[h] * a.scale;
                    }
                    wave_lds_sync();  // xq / qp are rewritten by the next query
                }
            }
            wave_lds_sync();
        }
    } else if (PHASE == 1) {
        // =========================== B: keys, scores, softmax, xbar =============================
        float *keys = wbase;
        float *qt = keys + a.K * KS;
        float *pb = qt + HP * CG;
        float *krel = pb + a.K * HP;
        int *krow = reinterpret_cast<int *>(krel + 3 * a.K);
        float wp[6], bpv;
#pragma unroll
        for (int t = 0; t < 6; ++t) wp[t] = a.Wp[(size_t)(a.c0 + cl) * 6 + t];
        bpv = a.bp[a.c0 + cl];
        const bool two_heads = a.K <= 32;  // score pass: lane = key + 32 * (head & 1)
        const int heads = a.heads;
        for (int wi = wfirst; wi < n_act; wi += wstep) {
            const int w = a.perm[wi];
            const float4 wc = a.wcentre[w];
            const float posc = bpv + wp[3] * wc.x + wp[4] * wc.y + wp[5] * wc.z;
            int nkv = 0, nqv = 0;
            {
                const float4 km = lane < a.K ? a.kmeta[(size_t)w * a.K + lane] : none4;
                const int row = __builtin_bit_cast(int, km.w);
                const bool ok = row >= 0;
                const unsigned long long m = __ballot(ok);
                if (ok) {
                    const int p = __popcll(m & ((1ull << lane) - 1ull));
                    krow[p] = row;
                    krel[3 * p + 0] = km.x;
                    krel[3 * p + 1] = km.y;
                    krel[3 * p + 2] = km.z;
                }
                nkv = __popcll(m);
            }
            for (int q0 = 0; q0 < a.nq; q0 += MSSVT_WAVE) {  // only the count is needed here
                const int qi = q0 + lane;
                const float4 qm = qi < a.nq ? a.qmeta[(size_t)w * a.nq + qi] : none4;
                nqv += __popcll(__ballot(__builtin_bit_cast(int, qm.w) >= 0));
            }
            wave_lds_sync();
            // key tokens: LN'd feature slice + positional embedding -> LDS, 16 row loads in flight
            for (int jb = 0; jb < nkv; jb += 16) {
                float val[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) val[u] = a.xhat[(size_t)krow[min(jb + u, nkv - 1)] * a.C + a.c0 + cl];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int jj = jb + u, jc = min(jj, nkv - 1);
                    const float pos = fmaxf(posc + wp[0] * krel[3 * jc] + wp[1] * krel[3 * jc + 1] + wp[2] * krel[3 * jc + 2], 0.0f);
                    if (jj < nkv && act) keys[jj * KS + lane] = val[u] + pos;
                }
            }
            const size_t qbase = (size_t)a.q_off[w];
            for (int qq = 0; qq < nqv; ++qq) {
                float *qrowp = a.qbuf + (qbase + qq) * QROW;
                if (act) {
#pragma unroll
                    for (int h = 0; h < NH; ++h) qt[h * CG + lane] = qrowp[h * CG + lane];
                }
                wave_lds_sync();
                // scores + softmax   (lane = key, two heads side by side when K <= 32)
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
                            float4 kk[STEP], qv4[STEP];
#pragma unroll
                            for (int i = 0; i < STEP; ++i) {
                                kk[i] = kr[b + i];
                                qv4[i] = qh[b + i];
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < STEP; ++i) {
                                s0 = __builtin_fmaf(kk[i].x, qv4[i].x, s0);
                                s1 = __builtin_fmaf(kk[i].y, qv4[i].y, s1);
                                s2 = __builtin_fmaf(kk[i].z, qv4[i].z, s2);
                                s3 = __builtin_fmaf(kk[i].w, qv4[i].w, s3);
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
                // xbar_h = sum_k p_hk x_k   (lane = channel, all heads at once, 8 keys per step)
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
                        if (h < NH) qrowp[h * CG + lane] = acc[h];  // xbar replaces qt in place
                }
                wave_lds_sync();  // qt / pb are rewritten by the next query
            }
            wave_lds_sync();  // keys / lists are rewritten for the next window
        }
    } else {
        // =========================== C: v = Wv xbar + bv, out = Wo v + bo ========================
        float *xbar = wbase;        // [HP][KS]
        float *vb = xbar + HP * KS;  // [CG]
        int *qslot = reinterpret_cast<int *>(vb + CG);
        const int my_h = cl / HD;
        for (int wi = wfirst; wi < n_act; wi += wstep) {
            const int w = a.perm[wi];
            int nqv = 0;
            for (int q0 = 0; q0 < a.nq; q0 += MSSVT_WAVE) {
                const int qi = q0 + lane;
                const float4 qm = qi < a.nq ? a.qmeta[(size_t)w * a.nq + qi] : none4;
                const bool ok = __builtin_bit_cast(int, qm.w) >= 0;
                const unsigned long long m = __ballot(ok);
                if (ok) qslot[nqv + __popcll(m & ((1ull << lane) - 1ull))] = qi;
                nqv += __popcll(m);
            }
            wave_lds_sync();
            const size_t qbase = (size_t)a.q_off[w];
            for (int qq = 0; qq < nqv; ++qq) {
                const float *qrowp = a.qbuf + (qbase + qq) * QROW;
                if (act) {
#pragma unroll
                    for (int h = 0; h < NH; ++h) xbar[h * KS + lane] = qrowp[h * CG + lane];
                }
                wave_lds_sync();
                const float vbv = matvec4<CG>(W0, xbar + my_h * KS, cl, b0[cl]);  // lane = output channel o
                if (act) vb[lane] = vbv;
                wave_lds_sync();
                const float out = matvec4<CG>(W1, vb, cl, b1[cl]);
                if (act) a.attn[((size_t)w * a.nq + qslot[qq]) * a.C + a.c0 + lane] = out;
                wave_lds_sync();  // xbar / vb are rewritten by the next query
            }
            wave_lds_sync();
        }
    }
}

template <int CG, int HD, int HP, int PHASE>
static int launch_attn_phase(AttnArgs a, hipStream_t stream) {
    constexpr int KS = CG + 4;
    size_t fixed = 0;
    if (PHASE == 0) {
        a.wave_floats = 2 * CG + 4 * a.nq;
        fixed = (size_t)2 * CG * CG + 2 * CG;
    } else if (PHASE == 1) {
        a.wave_floats = a.K * KS + HP * CG + a.K * HP + 4 * a.K;
        fixed = 0;
    } else {
        a.wave_floats = HP * KS + CG + a.nq;
        fixed = (size_t)2 * CG * CG + 2 * CG;
    }
    a.wave_floats = (a.wave_floats + 3) & ~3;  // keep every wave's region 16-B aligned
    int waves = ATTN_MAX_WAVES;
    while (waves > 1 && (fixed + (size_t)waves * a.wave_floats) * 4 > 160 * 1024) --waves;
    const size_t lds_bytes = (fixed + (size_t)waves * a.wave_floats) * 4;
    if (lds_bytes > 160 * 1024) return MSSVT_E_TOOLARGE;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_attn_phase<CG, HD, HP, PHASE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    // persistent grid: as many workgroups per CU as LDS and the 32-wave limit admit (256 CUs on MI355X)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    int per_cu = (int)((160 * 1024) / (lds_bytes ? lds_bytes : 1));
    if (per_cu > 32 / waves) per_cu = 32 / waves;
    if (per_cu < 1) per_cu = 1;
    k_attn_phase<CG, HD, HP, PHASE><<<cus * per_cu, waves * MSSVT_WAVE, lds_bytes, stream>>>(a);
    return mssvt_launch_status();
}

template <int CG, int HD, int HP>
static int launch_block_attn(AttnArgs &a, hipStream_t stream) {
    int rc = launch_attn_phase<CG, HD, HP, 0>(a, stream);
    if (rc) return rc;
    rc = launch_attn_phase<CG, HD, HP, 1>(a, stream);
    if (rc) return rc;
    return launch_attn_phase<CG, HD, HP, 2>(a, stream);
}

extern "C" int mssvt_block_attention_group(
    int C, int c0, int Cg, int heads, int head_dim, float scale, int nq, int key_num_sample,
    const float *xhat, const int *num_active_dev, const int *perm, const int *q_off, const float *qmeta,
    const float *kmeta, const float *wcentre, const float *Wq, const float *bq, const float *Wkv,
    const float *bkv, const float *Wo, const float *bo, const float *Wpos, const float *bpos, float *qbuf,
    float *attn, void *stream) {
    if (!xhat || !num_active_dev || !perm || !q_off || !qmeta || !kmeta || !wcentre || !Wq || !bq || !Wkv ||
        !bkv || !Wo || !bo || !Wpos || !bpos || !qbuf || !attn || C <= 0 || Cg <= 0 || heads <= 0 ||
        head_dim <= 0 || nq <= 0 || key_num_sample <= 0)
        return MSSVT_E_BADARG;
    if (Cg != heads * head_dim || c0 < 0 || c0 + Cg > C) return MSSVT_E_BADARG;
    // one channel per lane, heads aligned to 4-float LDS vectors, <= 8 heads per group
    if (Cg > MSSVT_WAVE || key_num_sample > MSSVT_WAVE || (head_dim & 3) || heads > 8) return MSSVT_E_TOOLARGE;
    AttnArgs a;
    a.C = C; a.c0 = c0; a.heads = heads; a.hd = head_dim; a.scale = scale;
    a.nq = nq; a.K = key_num_sample;
    a.xhat = xhat; a.num_wins = num_active_dev; a.perm = perm; a.q_off = q_off;
    a.qmeta = reinterpret_cast<const float4 *>(qmeta);
    a.kmeta = reinterpret_cast<const float4 *>(kmeta);
    a.wcentre = reinterpret_cast<const float4 *>(wcentre);
    a.Wq = Wq; a.bq = bq; a.Wkv = Wkv; a.bkv = bkv; a.Wo = Wo; a.bo = bo; a.Wp = Wpos; a.bp = bpos;
    a.qbuf = qbuf;
    a.attn = attn;
    a.wave_floats = 0;
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

// cell centre in metres, one rounding per op like the reference's torch expression
// (ref: with_coords, mssvt_backbone.py:132-137)
__device__ __forceinline__ float centre_of(int idx, float cell, float lo) {
    return __fadd_rn(__fmul_rn(__fadd_rn((float)idx, 0.5f), cell), lo);
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
