// common.hip.h -- shared device helpers for libmssvt_hip (gfx950 / CDNA4 only).
//
// Wavefront = 64 lanes everywhere in this library.  No CUDA compatibility
// layer, no multi-backend macros: this code is written for MI355X directly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mssvt_hip.h"

#define MSSVT_WAVE 64
#define MSSVT_EMPTY (-1)  // ref: mssvt/src/ms_cuda_utils.h:9 (EMPTY_KEY)

// workspace header words (int32) written by the hash builders
#define WS_STATUS 0       // bit0: duplicate key seen, bit1: table overflow, bit2: window overflow
#define WS_HDR_INTS 4
#define ST_DUP 1
#define ST_TABLE_OVERFLOW 2
#define ST_WIN_OVERFLOW 4
#define ST_UNSORTED 8  // mssvt_level_setup_sorted: the voxel list is not strictly (b,x,y,z)-ascending / in-grid

static inline int mssvt_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MSSVT_OK : (int)e;
}

static inline int divup(long long a, long long b) { return (int)((a + b - 1) / b); }

// launch-only halves of entry points whose output / status words must be zero on entry: the public
// functions clear them with their own fill, mssvt_level_setup clears all of them with ONE
int mssvt_batch_counts_launch(const int *indices, int num_rows, int batch_size, int *counts, hipStream_t stream);
int mssvt_occupancy_columns_launch(const int *indices, int num_voxels, int batch_size, int x_max, int y_max, int z_max,
                                   unsigned long long *columns, hipStream_t stream);

__device__ __forceinline__ int lane_id() { return threadIdx.x & (MSSVT_WAVE - 1); }

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one -- observed, MI355X_MICROARCH.md; used for
// speed only, never for correctness), and every XCD has an L2 of its own.  Work whose neighbours in the work order read the same
// rows (voxel tiles that gather the attention rows of shared windows, windows that gather the key rows of shared cells)
// should be dealt so that an XCD's workgroups own a CONTIGUOUS run of it each round: logical block = (b % 8) (n / 8) + b / 8,
// a permutation of [0, n) when 8 divides n (else the identity).  MSSVT_XCD_REMAP=0 turns it off (A / B runs).
__device__ __forceinline__ int xcd_contiguous_block(int b, int n, int on) {
    return (on && (n & 7) == 0) ? (b & 7) * (n >> 3) + (b >> 3) : b;
}
static inline int mssvt_xcd_remap() {
    static const int on = getenv("MSSVT_XCD_REMAP") ? atoi(getenv("MSSVT_XCD_REMAP")) : 1;
    return on;
}

// Ordering point for LDS traffic between lanes of ONE wave (LDS operations of a
// wave complete in order; this only stops the compiler from moving them).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// all-reduce inside each 16-lane row with DPP (no LDS round trip), then across rows
#define DPP_MOV(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
__device__ __forceinline__ float row_max16(float v) {
    v = fmaxf(v, DPP_MOV(v, 0xB1));   // quad_perm [1,0,3,2]
    v = fmaxf(v, DPP_MOV(v, 0x4E));   // quad_perm [2,3,0,1]
    v = fmaxf(v, DPP_MOV(v, 0x141));  // row_half_mirror
    v = fmaxf(v, DPP_MOV(v, 0x140));  // row_mirror
    return v;
}
__device__ __forceinline__ float row_sum16(float v) {
    v += DPP_MOV(v, 0xB1);
    v += DPP_MOV(v, 0x4E);
    v += DPP_MOV(v, 0x141);
    v += DPP_MOV(v, 0x140);
    return v;
}
// value of lane (l ^ 16) / (l ^ 32) through the gfx950 row / half swaps (VALU, no LDS round trip as
// ds_bpermute would be): v_permlane16_swap exchanges the odd rows of its first operand with the even
// rows of its second, v_permlane32_swap the upper half of the first with the lower half of the second
__device__ __forceinline__ float lane_xor16(float v) {
    const unsigned int u = __builtin_bit_cast(unsigned int, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (lane_id() & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor32(float v) {
    const unsigned int u = __builtin_bit_cast(unsigned int, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (lane_id() & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float half_max(float v) { v = row_max16(v); return fmaxf(v, lane_xor16(v)); }
__device__ __forceinline__ float half_sum(float v) { v = row_sum16(v); return v + lane_xor16(v); }
__device__ __forceinline__ float wave_max(float v) { v = half_max(v); return fmaxf(v, lane_xor32(v)); }
__device__ __forceinline__ float wave_sum(float v) { v = half_sum(v); return v + lane_xor32(v); }
// wave-uniform reductions without LDS traffic: DPP inside the 16-lane rows, v_readlane across them
__device__ __forceinline__ float wave_max_uniform(float v) {
    v = row_max16(v);
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// component g (0..3) of (x, y, z, w), lane by lane, as three v_cndmask (the nested `g == 0 ? x : (g == 1 ? ...` form is
// compiled into exec-masked branches: ~30 instructions and 6 branches per use inside the attention kernels' window loops)
__device__ __forceinline__ float lane_pick4(int g, float x, float y, float z, float w) {
    float r = w;
    r = g == 2 ? z : r;
    r = g == 1 ? y : r;
    r = g == 0 ? x : r;
    return r;
}
#define DPP_MOV_U(v, ctrl) (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xF, 0xF, true)
__device__ __forceinline__ unsigned int wave_min_u32_uniform(unsigned int v) {
    unsigned int o;
    o = DPP_MOV_U(v, 0xB1); v = o < v ? o : v;
    o = DPP_MOV_U(v, 0x4E); v = o < v ? o : v;
    o = DPP_MOV_U(v, 0x141); v = o < v ? o : v;
    o = DPP_MOV_U(v, 0x140); v = o < v ? o : v;
    const unsigned int r0 = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned int)__builtin_amdgcn_readlane((int)v, 16),
                       r2 = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned int a = r0 < r1 ? r0 : r1, b = r2 < r3 ? r2 : r3;
    return a < b ? a : b;
}
__device__ __forceinline__ int wave_sum_i(int v) {
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    return v;
}


// ---------------------------------------------------------------------------
// Hash table slots: (key, value) int32 pairs, ref layout (B, H, 2).  A slot is
// manipulated as one 64-bit word: low half = key, high half = value.
// ---------------------------------------------------------------------------
typedef unsigned long long slot_t;
#define SLOT_EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ slot_t slot_pack(int key, int value) {
    return (slot_t)(uint32_t)key | ((slot_t)(uint32_t)value << 32);
}
__device__ __forceinline__ int slot_key(slot_t s) { return (int)(uint32_t)(s & 0xFFFFFFFFull); }
__device__ __forceinline__ int slot_val(slot_t s) { return (int)(uint32_t)(s >> 32); }

__device__ __forceinline__ slot_t slot_load(const slot_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Lookup.  ref: hash_table_find, mssvt/src/ms_sparse_attention_gpu.cu:43-64.
// Returns the value or MSSVT_EMPTY; *slot_out (optional) = slot index or -1.
__device__ __forceinline__ int table_find(int key, int hash_size, const slot_t *tab,
                                          int *slot_out = nullptr) {
    int h = key % hash_size;
    int prob = 0;
    for (;;) {
        slot_t s = tab[h];
        int k = slot_key(s);
        if (k == key) {
            if (slot_out) *slot_out = h;
            return slot_val(s);
        }
        if (k == MSSVT_EMPTY) break;
        h = h + 1 == hash_size ? 0 : h + 1;
        if (++prob >= hash_size) break;
    }
    if (slot_out) *slot_out = -1;
    return MSSVT_EMPTY;
}

// Deterministic insertion: the final layout equals SEQUENTIAL linear-probing
// insertion (ref: hash_table_insert, ms_sparse_attention_gpu.cu:22-41) of the
// keys in increasing `value` order, whatever order the lanes actually run in.
// Rule: a slot always ends up holding the highest-priority (= lowest value)
// entry that probed it; the displaced entry carries on probing.  Slot contents
// only ever move towards higher priority, so a stale read can only cause a
// retry, never a wrong decision.  REQUIRES: keys inserted through this function
// into one table are pairwise distinct (duplicates are resolved by the
// callers first); returns bit flags ST_DUP / ST_TABLE_OVERFLOW it observed.
__device__ __forceinline__ int table_insert_ordered(int key, int value, int hash_size,
                                                    slot_t *tab) {
    slot_t cur = slot_pack(key, value);
    int i = key % hash_size;
    int dist = 0;
    for (;;) {
        slot_t c = slot_load(tab + i);
        for (;;) {  // settle this slot
            if (c == SLOT_EMPTY) {
                slot_t prev = atomicCAS(tab + i, c, cur);
                if (prev == c) return 0;
                c = prev;
                continue;
            }
            if (slot_key(c) == slot_key(cur)) return ST_DUP;
            if ((uint32_t)slot_val(c) > (uint32_t)slot_val(cur)) {  // cur outranks the resident
                slot_t prev = atomicCAS(tab + i, c, cur);
                if (prev == c) {
                    cur = c;  // carry the displaced entry onwards
                    int home = slot_key(cur) % hash_size;
                    dist = i - home;
                    if (dist < 0) dist += hash_size;
                    break;
                }
                c = prev;
                continue;
            }
            break;  // resident outranks cur: move on
        }
        i = i + 1 == hash_size ? 0 : i + 1;
        if (++dist >= hash_size) return ST_TABLE_OVERFLOW;  // ref :39 silent drop
    }
}

// Order-agnostic insertion keeping, per key, the MINIMUM value seen (used to
// find first occurrences).  Returns the slot index or -1 when the table is full.
__device__ __forceinline__ int table_insert_min(int key, int value, int hash_size, slot_t *tab) {
    int *words = reinterpret_cast<int *>(tab);
    int h = key % hash_size;
    int prob = 0;
    for (;;) {
        int prev = atomicCAS(words + 2 * h, MSSVT_EMPTY, key);
        if (prev == MSSVT_EMPTY || prev == key) {
            atomicMin(reinterpret_cast<unsigned int *>(words + 2 * h + 1), (unsigned int)value);
            return h;
        }
        h = h + 1 == hash_size ? 0 : h + 1;
        if (++prob >= hash_size) return -1;
    }
}

// ---- LayerNorm over the last dimension of (n, C = 4 LPR) rows: workgroup `block` of 256 threads normalises 4 * (64 / LPR)
// rows, LPR lanes (a float4 each) per row (k_layer_norm, rowops.hip; the fill launch of frame.hip)
template <int LPR>
__device__ __forceinline__ void layer_norm_rows(const float *x, int n, const float *w, const float *b, float eps, float *y,
                                                unsigned int block) {
    constexpr int C = LPR * 4, RPW = MSSVT_WAVE / LPR;
    const int lane = lane_id();
    const size_t row = ((size_t)block * 4 + threadIdx.x / MSSVT_WAVE) * RPW + lane / LPR;
    const int col = (lane % LPR) * 4;
    const bool live = row < (size_t)n;
    const float4 v = live ? *reinterpret_cast<const float4 *>(x + row * C + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    const float m = s * (1.0f / C);
    const float dx = v.x - m, dy = v.y - m, dz = v.z - m, dw = v.w - m;
    float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) q += __shfl_xor(q, off);
    const float rs = rsqrtf(q * (1.0f / C) + eps);
    const float4 g4 = *reinterpret_cast<const float4 *>(w + col);
    const float4 b4 = *reinterpret_cast<const float4 *>(b + col);
    if (live)
        *reinterpret_cast<float4 *>(y + row * C + col) =
            make_float4(dx * rs * g4.x + b4.x, dy * rs * g4.y + b4.y, dz * rs * g4.z + b4.z, dw * rs * g4.w + b4.w);
}
