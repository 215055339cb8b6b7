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
//   keys[K][Cg+1] (row stride Cg+1: conflict-free both by row and by column),
//   qt[heads][Cg], pb[heads][K], xbar[heads][Cg+1], krow[K].
#include "common.hip.h"

#define ATTN_MAX_WAVES 8

struct AttnArgs {
    int C, c0, Cg, heads, hd;
    float scale;
    int nq, K;
    const float *xhat;
    const int *indices, *win_ind, *num_wins, *win_vstart, *q_ind, *k_ind;
    const unsigned char *k_mask;
    float vsx, vsy, vsz, minx, miny, minz, wsx, wsy, wsz;
    const float *Wq, *bq, *Wkv, *bkv, *Wo, *bo, *Wp, *bp;
    float *attn;
    int wave_floats;
};

// cell centre in metres, one rounding per op like the reference's torch expression
// (ref: with_coords, mssvt_backbone.py:132-137)
__device__ __forceinline__ float centre_of(int idx, float cell, float lo) {
    return __fadd_rn(__fmul_rn(__fadd_rn((float)idx, 0.5f), cell), lo);
}

__device__ __forceinline__ float bcast(float v, int src_lane) {  // src_lane wave-uniform
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

__global__ void __launch_bounds__(ATTN_MAX_WAVES *MSSVT_WAVE) k_block_attn(AttnArgs a) {
    extern __shared__ float lds[];
    const int Cg = a.Cg, Cg2 = Cg * Cg, ks = Cg + 1;
    float *WqT = lds, *Wk = WqT + Cg2, *WvT = Wk + Cg2, *WoT = WvT + Cg2;
    float *bq = WoT + Cg2, *bv = bq + Cg, *bo = bv + Cg;
    const int nwaves = blockDim.x / MSSVT_WAVE, wv = threadIdx.x / MSSVT_WAVE, lane = lane_id();
    float *wbase = bo + Cg + (size_t)wv * a.wave_floats;
    float *keys = wbase;
    float *qt = keys + a.K * ks;
    float *pb = qt + a.heads * Cg;
    float *xbar = pb + a.heads * a.K;
    int *krow = reinterpret_cast<int *>(xbar + a.heads * ks);

    // ---- stage this group's weights (once per persistent workgroup) ----------------
    for (int e = threadIdx.x; e < Cg2; e += blockDim.x) {
        const int o = e / Cg, i = e % Cg;
        WqT[i * Cg + o] = a.Wq[e];
        Wk[e] = a.Wkv[e];                  // rows [0,Cg) of to_kvs: K projection, [o][i]
        WvT[i * Cg + o] = a.Wkv[Cg2 + e];  // rows [Cg,2Cg): V projection
        WoT[i * Cg + o] = a.Wo[e];
    }
    for (int e = threadIdx.x; e < Cg; e += blockDim.x) {
        bq[e] = a.bq[e];
        bv[e] = a.bkv[Cg + e];
        bo[e] = a.bo[e];
    }
    __syncthreads();

    const bool act = lane < Cg;
    const int cl = act ? lane : 0;
    float wp[6], bpv;  // positional MLP row of this lane's channel (ref pos_proj.0: (C,6,1))
#pragma unroll
    for (int t = 0; t < 6; ++t) wp[t] = a.Wp[(size_t)(a.c0 + cl) * 6 + t];
    bpv = a.bp[a.c0 + cl];
    const int nw = *a.num_wins;
    const bool two_heads = a.K <= 32;  // score pass: lane = key + 32 * (head & 1)
    const int my_h = act ? cl / a.hd : 0;

    for (int w = blockIdx.x * nwaves + wv; w < nw; w += gridDim.x * nwaves) {
        const int4 wi = reinterpret_cast<const int4 *>(a.win_ind)[w];  // [b,wz,wy,wx]
        const int vstart = a.win_vstart[w];
        const float cxm = centre_of(wi.w, a.wsx, a.minx), cym = centre_of(wi.z, a.wsy, a.miny),
                    czm = centre_of(wi.y, a.wsz, a.minz);
        // ---- unmasked keys -> compact row list ------------------------------------
        int nkv = 0;
        for (int j0 = 0; j0 < a.K; j0 += MSSVT_WAVE) {
            const int j = j0 + lane;
            const bool ok = j < a.K && a.k_mask[(size_t)w * a.K + j] == 0;
            const unsigned long long m = __ballot(ok);
            if (ok) krow[nkv + __popcll(m & ((1ull << lane) - 1ull))] = vstart + a.k_ind[(size_t)w * a.K + j];
            nkv += __popcll(m);
        }
        wave_lds_sync();
        // ---- key tokens: LN'd feature slice + positional embedding -> LDS ------------
#pragma unroll 4
        for (int jj = 0; jj < nkv; ++jj) {
            const int row = krow[jj];
            const int4 vi = reinterpret_cast<const int4 *>(a.indices)[row];
            const float rx = centre_of(vi.w, a.vsx, a.minx) - cxm, ry = centre_of(vi.z, a.vsy, a.miny) - cym,
                        rz = centre_of(vi.y, a.vsz, a.minz) - czm;
            float pos = bpv + wp[0] * rx + wp[1] * ry + wp[2] * rz + wp[3] * cxm + wp[4] * cym + wp[5] * czm;
            pos = fmaxf(pos, 0.0f);
            if (act) keys[jj * ks + lane] = a.xhat[(size_t)row * a.C + a.c0 + lane] + pos;
        }
        wave_lds_sync();
        // ---- queries -------------------------------------------------------------------
        for (int qi = 0; qi < a.nq; ++qi) {
            const int qid = a.q_ind[(size_t)w * a.nq + qi];
            if (qid < 0) continue;  // wave-uniform; K3 lists are front-packed but stay general
            const int row = vstart + qid;
            const int4 vi = reinterpret_cast<const int4 *>(a.indices)[row];
            const float rx = centre_of(vi.w, a.vsx, a.minx) - cxm, ry = centre_of(vi.z, a.vsy, a.miny) - cym,
                        rz = centre_of(vi.y, a.vsz, a.minz) - czm;
            float pos = bpv + wp[0] * rx + wp[1] * ry + wp[2] * rz + wp[3] * cxm + wp[4] * cym + wp[5] * czm;
            const float xq = act ? a.xhat[(size_t)row * a.C + a.c0 + lane] + fmaxf(pos, 0.0f) : 0.0f;
            // q' = Wq xq + bq        (lane = output channel)
            float qp = act ? bq[lane] : 0.0f;
            for (int i = 0; i < Cg; ++i) qp = __builtin_fmaf(WqT[i * Cg + cl], bcast(xq, i), qp);
            // qt_h = scale * Wk_h^T q'_h   (lane = input channel)
            for (int h = 0; h < a.heads; ++h) {
                float acc = 0.0f;
                for (int d = 0; d < a.hd; ++d) {
                    const int o = h * a.hd + d;
                    acc = __builtin_fmaf(Wk[o * Cg + cl], bcast(qp, o), acc);
                }
                if (act) qt[h * Cg + lane] = acc * a.scale;
            }
            wave_lds_sync();
            // scores + softmax       (lane = key, two heads side by side when K <= 32)
            const int j = two_heads ? (lane & 31) : lane;
            const int npass = two_heads ? (a.heads + 1) / 2 : a.heads;
            for (int p = 0; p < npass; ++p) {
                const int h = two_heads ? 2 * p + (lane >> 5) : p;
                const bool on = j < nkv && h < a.heads;
                float s = -INFINITY;
                if (on) {
                    s = 0.0f;
                    const float *kr = keys + j * ks, *qh = qt + h * Cg;
                    for (int c = 0; c < Cg; ++c) s = __builtin_fmaf(qh[c], kr[c], s);
                }
                float mx = s;
                for (int off = two_heads ? 16 : 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
                const float e = on ? expf(s - mx) : 0.0f;
                float sum = e;
                for (int off = two_heads ? 16 : 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
                if (on) pb[h * a.K + j] = e / sum;
            }
            wave_lds_sync();
            // xbar_h = sum_k p_hk x_k   (lane = channel)
            for (int h = 0; h < a.heads; ++h) {
                float acc = 0.0f;
                for (int jj = 0; jj < nkv; ++jj) acc = __builtin_fmaf(pb[h * a.K + jj], keys[jj * ks + cl], acc);
                if (act) xbar[h * ks + lane] = acc;
            }
            wave_lds_sync();
            // v = Wv xbar_h(o) + bv      (lane = output channel o, head of o = o / hd)
            float vb = act ? bv[lane] : 0.0f;
            {
                const float *xb = xbar + my_h * ks;
                for (int i = 0; i < Cg; ++i) vb = __builtin_fmaf(WvT[i * Cg + cl], xb[i], vb);
            }
            // out = Wo v + bo
            float out = act ? bo[lane] : 0.0f;
            for (int i = 0; i < Cg; ++i) out = __builtin_fmaf(WoT[i * Cg + cl], bcast(vb, i), out);
            if (act) a.attn[((size_t)w * a.nq + qi) * a.C + a.c0 + lane] = out;
            wave_lds_sync();  // qt / pb / xbar are reused by the next query
        }
    }
}

extern "C" int mssvt_block_attention_group(
    int C, int c0, int Cg, int heads, int head_dim, float scale, int nq, int key_num_sample,
    const float *xhat, const int *indices, const int *win_ind, const int *num_wins_dev,
    const int *win_vstart, const int *q_ind, const int *k_ind, const unsigned char *k_mask,
    const float *host_voxel_size3, const float *host_range_min3, const float *host_win_size3, const float *Wq,
    const float *bq, const float *Wkv, const float *bkv, const float *Wo, const float *bo,
    const float *Wpos, const float *bpos, float *attn, void *stream) {
    if (!xhat || !indices || !win_ind || !num_wins_dev || !win_vstart || !q_ind || !k_ind || !k_mask ||
        !host_voxel_size3 || !host_range_min3 || !host_win_size3 || !Wq || !bq || !Wkv || !bkv || !Wo || !bo || !Wpos ||
        !bpos || !attn || C <= 0 || Cg <= 0 || heads <= 0 || head_dim <= 0 || nq <= 0 || key_num_sample <= 0)
        return MSSVT_E_BADARG;
    if (Cg != heads * head_dim || c0 < 0 || c0 + Cg > C) return MSSVT_E_BADARG;
    if (Cg > MSSVT_WAVE || key_num_sample > MSSVT_WAVE) return MSSVT_E_TOOLARGE;  // v1: one channel per lane
    AttnArgs a;
    a.C = C; a.c0 = c0; a.Cg = Cg; a.heads = heads; a.hd = head_dim; a.scale = scale;
    a.nq = nq; a.K = key_num_sample;
    a.xhat = xhat; a.indices = indices; a.win_ind = win_ind; a.num_wins = num_wins_dev;
    a.win_vstart = win_vstart; a.q_ind = q_ind; a.k_ind = k_ind; a.k_mask = k_mask;
    a.vsx = host_voxel_size3[0]; a.vsy = host_voxel_size3[1]; a.vsz = host_voxel_size3[2];
    a.minx = host_range_min3[0]; a.miny = host_range_min3[1]; a.minz = host_range_min3[2];
    a.wsx = host_win_size3[0]; a.wsy = host_win_size3[1]; a.wsz = host_win_size3[2];
    a.Wq = Wq; a.bq = bq; a.Wkv = Wkv; a.bkv = bkv; a.Wo = Wo; a.bo = bo; a.Wp = Wpos; a.bp = bpos;
    a.attn = attn;
    const int ks = Cg + 1;
    a.wave_floats = key_num_sample * ks + heads * Cg + heads * key_num_sample + heads * ks + key_num_sample;
    const size_t fixed = (size_t)4 * Cg * Cg + 3 * Cg;
    int waves = ATTN_MAX_WAVES;
    while (waves > 1 && (fixed + (size_t)waves * a.wave_floats) * 4 > 160 * 1024) --waves;
    const size_t lds_bytes = (fixed + (size_t)waves * a.wave_floats) * 4;
    if (lds_bytes > 160 * 1024) return MSSVT_E_TOOLARGE;
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_block_attn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return (int)e;
    }
    // persistent grid: one workgroup per CU (LDS-bound residency), 256 CUs on MI355X
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int per_cu = lds_bytes * 2 <= 160 * 1024 ? 2 : 1;
    k_block_attn<<<cus * per_cu, waves * MSSVT_WAVE, lds_bytes, (hipStream_t)stream>>>(a);
    return mssvt_launch_status();
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
    int grid = divup(win_capacity, SC_WPB);
    if (grid > 4096) grid = 4096;  // grid-stride over the windows actually present
    k_block_scatter<<<grid, SC_WPB * MSSVT_WAVE, 0, (hipStream_t)stream>>>(a);
    return mssvt_launch_status();
}
