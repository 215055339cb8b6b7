// pair_attn.hip -- softmax attention of every query of a window against that window's keys on COMPACT rows, forward and
// backward (training path; the arithmetic of ref mssvt_utils.py:131-149 per head group).
//
//   window w:  queries = rows [q_off[w], q_off[w] + q_cnt[w]) of q (R, cg)  (already scaled),
//              keys    = rows [k_off[w], k_off[w] + k_cnt[w]) of kv (Kn, 2 cg) = [K | V]
//   O[i] = sum_j softmax_j(q_i . k_j per head) v_j         lse[i][h] = log sum_j exp(q_i . k_j)
//
// One wave per window, lane = channel (two channels per lane for 64 < cg <= 128): a K / V row is one coalesced 256-byte
// load, the per-head dot product a DPP sum over the head's lanes, softmax online (no (pairs, heads) score tensor in HBM).
// Eight queries are held in registers per pass over the keys.  Backward recomputes the probabilities from lse; every dq
// row and every dkv row belongs to exactly one wave and is accumulated in a fixed order (queries ascending inside a
// window): no atomics, bit-identical from run to run.  This replaces ~25 framework launches per head group (gather of the
// pairs' K/V rows, repeat_interleave, segment max/sum, their backward) and their (pairs, 2 cg) intermediates.
#include "common.hip.h"

#define PA_WAVES 4
#define PA_QB 8  // queries per pass

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

template <int CPL, int HD>
__global__ void __launch_bounds__(PA_WAVES *MSSVT_WAVE) k_pair_attn_fwd(int nw, int cg, int heads, const int *q_off, const int *q_cnt,
                                                                         const int *k_off, const int *k_cnt, const float *q,
                                                                         const float *kv, float *O, float *lse) {
    const int w = blockIdx.x * PA_WAVES + threadIdx.x / MSSVT_WAVE;
    if (w >= nw) return;
    const int lane = lane_id();
    const int qo = __builtin_amdgcn_readfirstlane(q_off[w]), nq = __builtin_amdgcn_readfirstlane(q_cnt[w]);
    const int ko = __builtin_amdgcn_readfirstlane(k_off[w]), nk = __builtin_amdgcn_readfirstlane(k_cnt[w]);
    bool act[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) act[r] = lane + 64 * r < cg;
    for (int i0 = 0; i0 < nq; i0 += PA_QB) {
        float qv[PA_QB][CPL], m[PA_QB][CPL], l[PA_QB][CPL], o[PA_QB][CPL];
#pragma unroll
        for (int b = 0; b < PA_QB; ++b)
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                qv[b][r] = (i0 + b < nq && act[r]) ? q[(size_t)(qo + i0 + b) * cg + lane + 64 * r] : 0.f;
                m[b][r] = -INFINITY; l[b][r] = 0.f; o[b][r] = 0.f;
            }
        for (int j = 0; j < nk; ++j) {
            float kk[CPL], vv[CPL];
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                const float *row = kv + (size_t)(ko + j) * 2 * cg + lane + 64 * r;
                kk[r] = act[r] ? row[0] : 0.f;
                vv[r] = act[r] ? row[cg] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < PA_QB; ++b) {
                if (i0 + b >= nq) break;  // wave-uniform
#pragma unroll
                for (int r = 0; r < CPL; ++r) {
                    const float s = head_sum<HD>(qv[b][r] * kk[r]);
                    const float mn = fmaxf(m[b][r], s);
                    const float corr = expf(m[b][r] - mn), p = expf(s - mn);
                    l[b][r] = l[b][r] * corr + p;
                    o[b][r] = o[b][r] * corr + p * vv[r];
                    m[b][r] = mn;
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

template <int CPL, int HD>
__global__ void __launch_bounds__(PA_WAVES *MSSVT_WAVE) k_pair_attn_bwd(int nw, int cg, int heads, const int *q_off, const int *q_cnt,
                                                                         const int *k_off, const int *k_cnt, const float *q,
                                                                         const float *kv, const float *O, const float *lse,
                                                                         const float *dO, float *dq, float *dkv) {
    const int w = blockIdx.x * PA_WAVES + threadIdx.x / MSSVT_WAVE;
    if (w >= nw) return;
    const int lane = lane_id();
    const int qo = __builtin_amdgcn_readfirstlane(q_off[w]), nq = __builtin_amdgcn_readfirstlane(q_cnt[w]);
    const int ko = __builtin_amdgcn_readfirstlane(k_off[w]), nk = __builtin_amdgcn_readfirstlane(k_cnt[w]);
    bool act[CPL];
#pragma unroll
    for (int r = 0; r < CPL; ++r) act[r] = lane + 64 * r < cg;
    if (nq == 0) {  // keys nobody looked at: zero gradient rows
        for (int j = 0; j < nk; ++j)
#pragma unroll
            for (int r = 0; r < CPL; ++r)
                if (act[r]) {
                    float *row = dkv + (size_t)(ko + j) * 2 * cg + lane + 64 * r;
                    row[0] = 0.f;
                    row[cg] = 0.f;
                }
        return;
    }
    for (int i0 = 0; i0 < nq; i0 += PA_QB) {
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
        for (int j = 0; j < nk; ++j) {
            float kk[CPL], vv[CPL], dk[CPL], dv[CPL];
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                const float *row = kv + (size_t)(ko + j) * 2 * cg + lane + 64 * r;
                kk[r] = act[r] ? row[0] : 0.f;
                vv[r] = act[r] ? row[cg] : 0.f;
                dk[r] = 0.f; dv[r] = 0.f;
            }
#pragma unroll
            for (int b = 0; b < PA_QB; ++b) {
                if (i0 + b >= nq) break;  // wave-uniform
#pragma unroll
                for (int r = 0; r < CPL; ++r) {
                    const float s = head_sum<HD>(qv[b][r] * kk[r]);
                    const float p = expf(s - ls[b][r]);
                    const float dp = head_sum<HD>(dov[b][r] * vv[r]);
                    const float ds = p * (dp - delta[b][r]);
                    dv[r] += p * dov[b][r];
                    dk[r] += ds * qv[b][r];
                    dqa[b][r] += ds * kk[r];
                }
            }
#pragma unroll
            for (int r = 0; r < CPL; ++r)
                if (act[r]) {
                    float *row = dkv + (size_t)(ko + j) * 2 * cg + lane + 64 * r;
                    if (i0 == 0) { row[0] = dk[r]; row[cg] = dv[r]; }
                    else { row[0] += dk[r]; row[cg] += dv[r]; }  // this wave's own rows: earlier passes, in order
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

static int pa_check(int nw, int cg, int heads, int hd) {
    if (nw < 0 || cg <= 0 || cg > 128 || heads <= 0 || heads * hd != cg) return MSSVT_E_BADARG;
    if (hd != 4 && hd != 8 && hd != 16 && hd != 32 && hd != 64) return MSSVT_E_BADARG;
    return 0;
}

#define PA_DISPATCH(KERNEL, ...)                                                                          \
    do {                                                                                                  \
        const dim3 grid(divup(nw, PA_WAVES)), block(PA_WAVES *MSSVT_WAVE);                                 \
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
    PA_DISPATCH(k_pair_attn_bwd, nw, cg, heads, q_off, q_cnt, k_off, k_cnt, q, kv, O, lse, dO, dq, dkv);
    return mssvt_launch_status();
}
