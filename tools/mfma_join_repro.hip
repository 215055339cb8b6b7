// The join pattern of DESIGN 5.0000 item 2, reduced: a chain of v_mfma_f32_16x16x4_f32, a wave-uniform branch around a second
// chain, the first VALU read of the first chain's sums right behind the join (10 wait states required: 8 passes + 2).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -c tools/mfma_join_repro.hip -o /tmp/r.o
//   (then tools/mfma_hazard_check.py's check_text on `llvm-objdump -d` of the gfx950 code object)
// In THIS reduced form hipcc (ROCm 7.2) pads correctly: `s_nop 8` sits at the join, the skip path reaches the read after
// s_cbranch (1) + s_nop 8 (9) = 10 states.  In k_attn_kv (round-4 HEAD) the skip path additionally ran the zero-initialisation
// of the skipped chain's accumulator in a block of its own in front of the join,
//     v_mfma_f32_16x16x4_f32 v[38:41], v3, v37, v[38:41]
//     s_cbranch_vccnz  -> L                      ; 1
//     (4 more v_mfma ..., s_branch -> J)          ; fall-through path: 4 + 1 + s_nop 3 (4) = 9 + 1 = 10
//  L: v_mov_b32 v34..v37, 0                       ; 4
//  J: s_nop 3                                     ; 4  -> 9 wait states on the skip path
//     v_add_f32 v25, 0, v38                       ; first read of the chain's sums
// i.e. the padding was sized for the fall-through path.  The library no longer contains the pattern (tile 0 runs last and
// unconditionally); tests/test_mfma_hazard_cpu.py::test_checker_flags_a_read_behind_a_join... keeps the listing as a fixture.
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void repro(const float *a, const float *b, const int *flag, float *o) {
    const int l = threadIdx.x;
    f4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    const float a0 = a[l], a1 = a[l + 64], a2 = a[l + 128], a3 = a[l + 192], b0 = b[l], b1 = b[l + 64], b2 = b[l + 128], b3 = b[l + 192];
    const int used = __builtin_amdgcn_readfirstlane(flag[0]);
    s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, s0, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, s0, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, s0, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, s0, 0, 0, 0);
    if (used & 2) {  // wave-uniform: a scalar branch around the second chain
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b2, s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b3, s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b0, s1, 0, 0, 0);
    }
    o[l] = (s0[0] + 0.0f) + s0[1] + s1[0] + s1[1];  // first read of s0 right behind the join
}
