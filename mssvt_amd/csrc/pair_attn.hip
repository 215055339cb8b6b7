// pair_attn.hip -- softmax attention of every query of a window against that window's keys on COMPACT rows, forward and
// backward (training path; the arithmetic of ref mssvt_utils.py:131-149 per head group).
//
//   window w:  queries = rows [q_off[w], q_off[w] + q_cnt[w]) of q (R, cg)  (already scaled),
//              keys    = rows [k_off[w], k_off[w] + k_cnt[w]) of kv (Kn, 2 cg) = [K | V]
//   O[i] = sum_j softmax_j(q_i . k_j per head) v_j         lse[i][h] = log sum_j exp(q_i . k_j)
//
// Lane = channel (two channels per lane for 64 < cg <= 128): a K / V row is one coalesced 256-byte load, the per-head dot
// product a DPP sum over the head's lanes, softmax online (no (pairs, heads) score tensor in HBM).  A work item holds eight
// rows of one side in registers and streams the other side past them (rows loaded a chunk ahead of their use):
//   forward, dq : item = (window, 8 queries), streams the window's keys;
//   dK / dV     : item = (window, 8 keys),    streams the window's queries.
// A workgroup owns four consecutive windows and deals their items round-robin to its four waves: a window with 45 queries
// is six items on four waves, not one wave's 70 us tail (one wave per window: 82 / 94 us per launch, the heaviest window's
// time).  Backward recomputes the probabilities from lse; every dq row and every dkv row is written by exactly one item and
// accumulated in ascending row order: no atomics, bit-identical from run to run.  This replaces ~25 framework launches
// per head group (gather of the pairs' K/V rows, repeat_interleave, segment max/sum, their backward) and their
// (pairs, 2 cg) intermediates.
#include "common.hip.h"

#define PA_WAVES 4
#define PA_QB 8  // queries per pass
#define PA_KB 4  // key rows per chunk of loads
// rows [J0_, J0_ + PA_KB) of the window's keys (clamped to its last row: never used past nk, never out of bounds)
#define PA_LOAD_QUERIES(QN_, GN_, ON_, LN_, I0_)                                                    \
    if (nq > 0) {                                                                                  \
        _Pragma("unroll") for (int t_ = 0; t_ < PA_KB; ++t_) {                                     \
            const size_t row_ = (size_t)(qo + min((I0_) + t_, nq - 1));                            \
            _Pragma("unroll") for (int r_ = 0; r_ < CPL; ++r_) {                                   \
                const int c_ = lane + 64 * r_;                                                     \
                QN_[t_][r_] = act[r_] ? q[row_ * cg + c_] : 0.f;                                   \
                GN_[t_][r_] = act[r_] ? dO[row_ * cg + c_] : 0.f;                                  \
                ON_[t_][r_] = act[r_] ? O[row_ * cg + c_] : 0.f;                                   \
                LN_[t_][r_] = act[r_] ? lse[row_ * heads + c_ / HD] : 0.f;                         \
            }                                                                                      \
        }                                                                                          \
    }
#define PA_LOAD_CHUNK(KN_, VN_, J0_)                                                               \
    if (nk > 0) {                                                                                  \
        _Pragma("unroll") for (int t_ = 0; t_ < PA_KB; ++t_) {                                     \
            const int jj_ = min((J0_) + t_, nk - 1);                                               \
            _Pragma("unroll") for (int r_ = 0; r_ < CPL; ++r_) {                                   \
                const float *row_ = kv + (size_t)(ko + jj_) * 2 * cg + lane + 64 * r_;             \
                KN_[t_][r_] = act[r_] ? row_[0] : 0.f;                                             \
                VN_[t_][r_] = act[r_] ? row_[cg] : 0.f;                                            \
            }                                                                                      \
        }                                                                                          \
    }

// e^x as one multiply and the hardware exp2 (relative error ~ |x| 2^-24: the library expf is ~15 instructions, and with a
// lane per channel every lane of a head evaluates it -- two thirds of the kernels' issue slots went there)
__device__ __forceinline__ float pa_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

template <int HD>
__device__ __forceinline__ float head_sum(float v) {
    if (HD >= 2) v += DPP_MOV(v, 0xB1);   // quad_perm [1,0,3,2]
    if (HD >= 4) v += DPP_MOV(v, 0x4E);   // quad_perm [2,3,0,1]
    if (HD >= 8) v += DPP_MOV(v, 0x141);  // row_half_mirror
    if (HD >= 16) v += DPP_MOV(v, 0x140); // row_mirror
    if (HD >= 32) v += lane_xor16(v);
    if (HD >= 64) v += lane_xor32(v);
    return v;
}

// the four windows of a workgroup and the t-th of their work items (all wave-uniform)
struct PaWin {
    int qo[4], nq[4], ko[4], nk[4], cnt[4], total;
};
__device__ __forceinline__ PaWin pa_windows(int nw, const int *q_off, const int *q_cnt, const int *k_off, const int *k_cnt,
                                            bool by_keys) {
    PaWin W;
    W.total = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int w = blockIdx.x * 4 + a;
        const bool in = w < nw;
        W.qo[a] = in ? __builtin_amdgcn_readfirstlane(q_off[w]) : 0;
        W.nq[a] = in ? __builtin_amdgcn_readfirstlane(q_cnt[w]) : 0;
        W.ko[a] = in ? __builtin_amdgcn_readfirstlane(k_off[w]) : 0;
        W.nk[a] = in ? __builtin_amdgcn_readfirstlane(k_cnt[w]) : 0;
        W.cnt[a] = ((by_keys ? W.nk[a] : W.nq[a]) + PA_QB - 1) / PA_QB;
        W.total += W.cnt[a];
    }
    return W;
}
#define PA_ITEM(W_, T_, A_, LOCAL_)                                  \
    int A_ = 0, LOCAL_ = (T_);                                       \
    _Pragma("unroll") for (int a_ = 0; a_ < 3; ++a_)                  \
        if (A_ == a_ && LOCAL_ >= W_.cnt[a_]) { LOCAL_ -= W_.cnt[a_]; A_ = a_ + 1; }
#define PA_PICK(ARR_, A_) ((A_) == 0 ? ARR_[0] : (A_) == 1 ? ARR_[1] : (A_) == 2 ? ARR_[2] : ARR_[3])

template <int CPL, int HD>
__global__ void __launch_bounds__(PA_WAVES *MSSVT_WAVE) k_pair_attn_fwd(int nw, int cg, int heads, const int *q_off, const int *q_cnt,
                                                                         const int *k_off, const int *k_cnt, const float *q,
                                                                         const float *kv, float *O, float *lse) {
    const PaWin W = pa_windows(nw, q_off, q_cnt, k_off, k_cnt, false);
    const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    bool act[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) act[r] = lane + 64 * r < cg;
    for (int t = wv; t < W.total; t += PA_WAVES) {
        PA_ITEM(W, t, a, local);
        const int qo = PA_PICK(W.qo, a), nq = PA_PICK(W.nq, a), ko = PA_PICK(W.ko, a), nk = PA_PICK(W.nk, a), i0 = local * PA_QB;
        float qv[PA_QB][CPL], m[PA_QB][CPL], l[PA_QB][CPL], o[PA_QB][CPL];
#pragma unroll
        for (int b = 0; b < PA_QB; ++b)
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                qv[b][r] = (i0 + b < nq && act[r]) ? q[(size_t)(qo + i0 + b) * cg + lane + 64 * r] : 0.f;
                m[b][r] = -INFINITY; l[b][r] = 0.f; o[b][r] = 0.f;
            }
        float kn[PA_KB][CPL], vn[PA_KB][CPL];
        PA_LOAD_CHUNK(kn, vn, 0);
        for (int j0 = 0; j0 < nk; j0 += PA_KB) {
            float kc[PA_KB][CPL], vc[PA_KB][CPL];
#pragma unroll
            for (int u = 0; u < PA_KB; ++u)
#pragma unroll
                for (int r = 0; r < CPL; ++r) { kc[u][r] = kn[u][r]; vc[u][r] = vn[u][r]; }
            if (j0 + PA_KB < nk) PA_LOAD_CHUNK(kn, vn, j0 + PA_KB);
#pragma unroll
            for (int u = 0; u < PA_KB; ++u) {
                if (j0 + u >= nk) break;  // wave-uniform
#pragma unroll
                for (int b = 0; b < PA_QB; ++b) {
                    if (i0 + b >= nq) break;  // wave-uniform
#pragma unroll
                    for (int r = 0; r < CPL; ++r) {
                        const float s = head_sum<HD>(qv[b][r] * kc[u][r]);
                        const float mn = fmaxf(m[b][r], s);
                        const float corr = pa_exp(m[b][r] - mn), p = pa_exp(s - mn);
                        l[b][r] = l[b][r] * corr + p;
                        o[b][r] = o[b][r] * corr + p * vc[u][r];
                        m[b][r] = mn;
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < PA_QB; ++b) {
            if (i0 + b >= nq) break;
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                if (!act[r]) continue;
                const int c = lane + 64 * r;
                const size_t row = (size_t)(qo + i0 + b);
                O[row * cg + c] = nk > 0 ? o[b][r] / l[b][r] : 0.f;
                if (c % HD == 0) lse[row * heads + c / HD] = nk > 0 ? m[b][r] + logf(l[b][r]) : 0.f;
            }
        }
    }
}

// dq[i] = sum_j ds_ij k_j,  ds_ij = p_ij (dO_i . v_j - dO_i . O_i)   (items = 8 queries, keys streamed)
template <int CPL, int HD>
__global__ void __launch_bounds__(PA_WAVES *MSSVT_WAVE) k_pair_attn_dq(int nw, int cg, int heads, const int *q_off, const int *q_cnt,
                                                                        const int *k_off, const int *k_cnt, const float *q,
                                                                        const float *kv, const float *O, const float *lse,
                                                                        const float *dO, float *dq) {
    const PaWin W = pa_windows(nw, q_off, q_cnt, k_off, k_cnt, false);
    const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    bool act[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) act[r] = lane + 64 * r < cg;
    for (int t = wv; t < W.total; t += PA_WAVES) {
        PA_ITEM(W, t, a, local);
        const int qo = PA_PICK(W.qo, a), nq = PA_PICK(W.nq, a), ko = PA_PICK(W.ko, a), nk = PA_PICK(W.nk, a), i0 = local * PA_QB;
        float qv[PA_QB][CPL], dov[PA_QB][CPL], delta[PA_QB][CPL], ls[PA_QB][CPL], dqa[PA_QB][CPL];
#pragma unroll
        for (int b = 0; b < PA_QB; ++b)
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                const bool ok = i0 + b < nq && act[r];
                const size_t row = (size_t)(qo + i0 + b);
                const int c = lane + 64 * r;
                qv[b][r] = ok ? q[row * cg + c] : 0.f;
                dov[b][r] = ok ? dO[row * cg + c] : 0.f;
                const float ov = ok ? O[row * cg + c] : 0.f;
                ls[b][r] = ok ? lse[row * heads + c / HD] : 0.f;
                delta[b][r] = head_sum<HD>(dov[b][r] * ov);
                dqa[b][r] = 0.f;
            }
        float kn[PA_KB][CPL], vn[PA_KB][CPL];
        PA_LOAD_CHUNK(kn, vn, 0);
        for (int j0 = 0; j0 < nk; j0 += PA_KB) {
            float kc[PA_KB][CPL], vc[PA_KB][CPL];
#pragma unroll
            for (int u = 0; u < PA_KB; ++u)
#pragma unroll
                for (int r = 0; r < CPL; ++r) { kc[u][r] = kn[u][r]; vc[u][r] = vn[u][r]; }
            if (j0 + PA_KB < nk) PA_LOAD_CHUNK(kn, vn, j0 + PA_KB);
#pragma unroll
            for (int u = 0; u < PA_KB; ++u) {
                if (j0 + u >= nk) break;  // wave-uniform
#pragma unroll
                for (int b = 0; b < PA_QB; ++b) {
                    if (i0 + b >= nq) break;  // wave-uniform
#pragma unroll
                    for (int r = 0; r < CPL; ++r) {
                        const float s = head_sum<HD>(qv[b][r] * kc[u][r]);
                        const float p = pa_exp(s - ls[b][r]);
                        const float dp = head_sum<HD>(dov[b][r] * vc[u][r]);
                        dqa[b][r] += p * (dp - delta[b][r]) * kc[u][r];
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < PA_QB; ++b) {
            if (i0 + b >= nq) break;
#pragma unroll
            for (int r = 0; r < CPL; ++r)
                if (act[r]) dq[(size_t)(qo + i0 + b) * cg + lane + 64 * r] = dqa[b][r];
        }
    }
}

// dK[j] = sum_i ds_ij q_i,  dV[j] = sum_i p_ij dO_i   (items = 8 keys, queries streamed in ascending order)
template <int CPL, int HD>
__global__ void __launch_bounds__(PA_WAVES *MSSVT_WAVE) k_pair_attn_dkv(int nw, int cg, int heads, const int *q_off, const int *q_cnt,
                                                                         const int *k_off, const int *k_cnt, const float *q,
                                                                         const float *kv, const float *O, const float *lse,
                                                                         const float *dO, float *dkv) {
    const PaWin W = pa_windows(nw, q_off, q_cnt, k_off, k_cnt, true);
    const int lane = lane_id(), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / MSSVT_WAVE);
    bool act[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) act[r] = lane + 64 * r < cg;
    for (int t = wv; t < W.total; t += PA_WAVES) {
        PA_ITEM(W, t, a, local);
        const int qo = PA_PICK(W.qo, a), nq = PA_PICK(W.nq, a), ko = PA_PICK(W.ko, a), nk = PA_PICK(W.nk, a), j0 = local * PA_QB;
        float kk[PA_QB][CPL], vv[PA_QB][CPL], dk[PA_QB][CPL], dv[PA_QB][CPL];
#pragma unroll
        for (int b = 0; b < PA_QB; ++b)
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                const bool ok = j0 + b < nk && act[r];
                const float *row = kv + (size_t)(ko + j0 + b) * 2 * cg + lane + 64 * r;
                kk[b][r] = ok ? row[0] : 0.f;
                vv[b][r] = ok ? row[cg] : 0.f;
                dk[b][r] = 0.f; dv[b][r] = 0.f;
            }
        float qn[PA_KB][CPL], gn[PA_KB][CPL], on[PA_KB][CPL], ln[PA_KB][CPL];
        PA_LOAD_QUERIES(qn, gn, on, ln, 0);
        for (int i0 = 0; i0 < nq; i0 += PA_KB) {
            float qc[PA_KB][CPL], gc[PA_KB][CPL], dl[PA_KB][CPL], lc[PA_KB][CPL];
#pragma unroll
            for (int u = 0; u < PA_KB; ++u)
#pragma unroll
                for (int r = 0; r < CPL; ++r) {
                    qc[u][r] = qn[u][r]; gc[u][r] = gn[u][r]; lc[u][r] = ln[u][r];
                    dl[u][r] = gn[u][r] * on[u][r];
                }
            if (i0 + PA_KB < nq) PA_LOAD_QUERIES(qn, gn, on, ln, i0 + PA_KB);
#pragma unroll
            for (int u = 0; u < PA_KB; ++u) {
                if (i0 + u >= nq) break;  // wave-uniform
#pragma unroll
                for (int r = 0; r < CPL; ++r) dl[u][r] = head_sum<HD>(dl[u][r]);
#pragma unroll
                for (int b = 0; b < PA_QB; ++b) {
                    if (j0 + b >= nk) break;  // wave-uniform
#pragma unroll
                    for (int r = 0; r < CPL; ++r) {
                        const float s = head_sum<HD>(qc[u][r] * kk[b][r]);
                        const float p = pa_exp(s - lc[u][r]);
                        const float dp = head_sum<HD>(gc[u][r] * vv[b][r]);
                        dv[b][r] += p * gc[u][r];
                        dk[b][r] += p * (dp - dl[u][r]) * qc[u][r];
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < PA_QB; ++b) {
            if (j0 + b >= nk) break;
#pragma unroll
            for (int r = 0; r < CPL; ++r)
                if (act[r]) {
                    float *row = dkv + (size_t)(ko + j0 + b) * 2 * cg + lane + 64 * r;
                    row[0] = dk[b][r];
                    row[cg] = dv[b][r];
                }
        }
    }
}

static int pa_check(int nw, int cg, int heads, int hd) {
    if (nw < 0 || cg <= 0 || cg > 128 || heads <= 0 || heads * hd != cg) return MSSVT_E_BADARG;
    if (hd != 4 && hd != 8 && hd != 16 && hd != 32 && hd != 64) return MSSVT_E_BADARG;
    return 0;
}

#define PA_DISPATCH(KERNEL, ...)                                                                          \
    do {                                                                                                  \
        const dim3 grid(divup(nw, 4)), block(PA_WAVES *MSSVT_WAVE);                                 \
        if (cg <= 64) {                                                                                   \
            if (hd == 4) KERNEL<1, 4><<<grid, block, 0, st>>>(__VA_ARGS__);                               \
            else if (hd == 8) KERNEL<1, 8><<<grid, block, 0, st>>>(__VA_ARGS__);                          \
            else if (hd == 16) KERNEL<1, 16><<<grid, block, 0, st>>>(__VA_ARGS__);                        \
            else if (hd == 32) KERNEL<1, 32><<<grid, block, 0, st>>>(__VA_ARGS__);                        \
            else KERNEL<1, 64><<<grid, block, 0, st>>>(__VA_ARGS__);                                      \
        } else {                                                                                          \
            if (hd == 4) KERNEL<2, 4><<<grid, block, 0, st>>>(__VA_ARGS__);                               \
            else if (hd == 8) KERNEL<2, 8><<<grid, block, 0, st>>>(__VA_ARGS__);                          \
            else if (hd == 16) KERNEL<2, 16><<<grid, block, 0, st>>>(__VA_ARGS__);                        \
            else if (hd == 32) KERNEL<2, 32><<<grid, block, 0, st>>>(__VA_ARGS__);                        \
            else KERNEL<2, 64><<<grid, block, 0, st>>>(__VA_ARGS__);                                      \
        }                                                                                                 \
    } while (0)

extern "C" int mssvt_pair_attention_fwd(int nw, int cg, int heads, int hd, const int *q_off, const int *q_cnt, const int *k_off,
                                        const int *k_cnt, const float *q, const float *kv, float *O, float *lse, void *stream) {
    if (int e = pa_check(nw, cg, heads, hd)) return e;
    if (nw == 0) return 0;
    if (!q_off || !q_cnt || !k_off || !k_cnt) return MSSVT_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    PA_DISPATCH(k_pair_attn_fwd, nw, cg, heads, q_off, q_cnt, k_off, k_cnt, q, kv, O, lse);
    return mssvt_launch_status();
}

extern "C" int mssvt_pair_attention_bwd(int nw, int cg, int heads, int hd, const int *q_off, const int *q_cnt, const int *k_off,
                                        const int *k_cnt, const float *q, const float *kv, const float *O, const float *lse,
                                        const float *dO, float *dq, float *dkv, void *stream) {
    if (int e = pa_check(nw, cg, heads, hd)) return e;
    if (nw == 0) return 0;
    if (!q_off || !q_cnt || !k_off || !k_cnt) return MSSVT_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (dq) PA_DISPATCH(k_pair_attn_dq, nw, cg, heads, q_off, q_cnt, k_off, k_cnt, q, kv, O, lse, dO, dq);
    PA_DISPATCH(k_pair_attn_dkv, nw, cg, heads, q_off, q_cnt, k_off, k_cnt, q, kv, O, lse, dO, dkv);
    return mssvt_launch_status();
}
