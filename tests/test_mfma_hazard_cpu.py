"""Build-time guard: no instruction of the shipped gfx950 code touches the destination of a matrix instruction before
its result has landed, on ANY path of the control-flow graph (tools/mfma_hazard_check.py; DESIGN 5.000 item 2: a
build of k_cmp_ws once read its sums one term short where a branch joined behind the chain -- 1e-4 errors that the
feature tolerance cannot tell from noise).  Runs in the CPU-only container: hipcc cross-compiles, llvm-objdump reads
the code objects inside libmssvt_hip.so."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mfma_hazard_check as hz  # noqa: E402

LIB = os.path.join(ROOT, "mssvt_amd", "lib", "libmssvt_hip.so")
needs_llvm = pytest.mark.skipif(not os.path.exists(os.path.join(hz.LLVM_BIN, "llvm-objdump")), reason="no llvm-objdump")
needs_hipcc = pytest.mark.skipif(shutil.which(hz.HIPCC) is None and not os.path.exists(hz.HIPCC), reason="no hipcc")


def _line(addr, text, target=None):
    tail = " <k+0x%x>" % target if target is not None else ""
    return "\t%-58s // %012X: 00000000%s" % (text, addr, tail)


def _listing(lines):
    """An llvm-objdump style listing of one kernel `k` at address 0; lines = (text, branch target or None)."""
    out, addr = ["0000000000000000 <k>:"], 0
    for text, target in lines:
        out.append(_line(addr, text, target))
        addr += 4
    return "\n".join(out) + "\n"


F32 = "v_mfma_f32_16x16x4_f32"
H16 = "v_mfma_f32_16x16x32_f16"


def test_checker_flags_a_read_behind_a_join_and_accepts_the_padded_form():
    # a wave-uniform branch skips the tail of the chain: the skip path reaches the first read after 1 + 4 + 4 = 9 states
    short = [(F32 + " v[38:41], v1, v35, v[38:41]", None), ("s_cbranch_vccnz 2", 4 * 4),  # -> the v_mov block
             (F32 + " v[68:71], v10, v34, 0", None), ("s_branch 0", 4 * 8),
             ("v_mov_b32_e32 v34, 0", None), ("v_mov_b32_e32 v35, 0", None), ("v_mov_b32_e32 v36, 0", None),
             ("v_mov_b32_e32 v37, 0", None),
             ("s_nop 3", None), ("v_add_f32_e32 v25, 0, v38", None), ("s_endpgm", None)]
    v, nf, nm = hz.check_text(_listing(short))
    assert (nf, nm) == (1, 2)
    assert [(x.kind, x.have, x.need) for x in v] == [(hz.VALU, 9, 10)]
    padded = list(short)
    padded[8] = ("s_nop 4", None)
    assert hz.check_text(_listing(padded))[0] == []


def test_checker_knows_chains_the_matrix_pipe_and_overwrites():
    # an accumulation chain needs nothing; a later matrix instruction cannot issue before the pipe is free (8 passes), so
    # two more states behind it are enough -- the strict count (one state per instruction) still reports that site
    chain = [(F32 + " v[0:3], v8, v9, v[0:3]", None), (F32 + " v[0:3], v10, v11, v[0:3]", None), ("s_nop 9", None),
             ("v_add_f32_e32 v4, v0, v1", None), ("s_endpgm", None)]
    assert hz.check_text(_listing(chain))[0] == []
    pipe = [(F32 + " v[0:3], v8, v9, v[0:3]", None), (F32 + " v[4:7], v10, v11, v[4:7]", None), ("s_nop 0", None),
            ("v_mov_b32_e32 v0, 0", None), ("s_nop 9", None), ("s_endpgm", None)]
    assert hz.check_text(_listing(pipe))[0] == []
    strict = hz.check_text(_listing(pipe), strict=True)[0]
    assert [(x.kind, x.have, x.need) for x in strict] == [(hz.VALU, 2, 10)]
    # ... but an overwrite (or a store) two states behind the LAST matrix instruction is an error in both models
    waw = [(F32 + " v[0:3], v8, v9, v[0:3]", None), ("s_nop 0", None), ("v_mov_b32_e32 v0, 0", None), ("s_nop 9", None),
           ("global_store_dwordx4 v[10:11], v[0:3], off", None), ("s_endpgm", None)]
    assert [(x.kind, x.have, x.need) for x in hz.check_text(_listing(waw))[0]] == [(hz.VALU, 1, 10)]
    store = [(H16 + " v[0:3], v[8:11], v[12:15], v[0:3]", None), ("s_nop 5", None),
             ("global_store_dwordx4 v[20:21], v[0:3], off", None), ("s_endpgm", None)]
    assert [(x.kind, x.have, x.need) for x in hz.check_text(_listing(store))[0]] == [(hz.MEM, 6, 8)]
    # a loop back-edge is a path too
    loop = [("v_mov_b32_e32 v0, 0", None), (F32 + " v[0:3], v8, v9, 0", None), ("s_cbranch_scc1 65534", 0), ("s_endpgm", None)]
    assert [(x.kind, x.have) for x in hz.check_text(_listing(loop))[0]] == [(hz.VALU, 1)]
    # an instruction the table does not know fails loudly
    unknown = [("v_mfma_f32_32x32x4_xf32 v[0:15], v[16:17], v[18:19], v[0:15]", None), ("s_endpgm", None)]
    assert hz.check_text(_listing(unknown))[0][0].kind == "table"


@needs_hipcc
def test_the_wait_state_table_is_what_this_compiler_pads_in_straight_line_code():
    """One-MFMA kernels compiled with the library's compiler: the s_nop count it leaves in front of the first VALU read, the
    first store and a dependent matrix instruction IS the rule the checker applies (a toolchain that changes it is seen)."""
    cal = hz.calibrate()
    assert len(cal) == 12, cal
    for (mnem, kind), have in cal.items():
        need = hz.required(mnem, kind)
        assert have == need, (mnem, kind, have, need)


@needs_llvm
def test_library_has_no_mfma_result_hazard():
    assert os.path.exists(LIB), "build first: python -m mssvt_amd.build"
    v, kernels, mfmas = hz.check_library(LIB)
    assert kernels > 300 and mfmas > 10000  # the code objects were found and read
    assert not v, "\n".join("%s: %s @%x -> %s @%x: %d of %d wait states" %
                            (x.func, x.mfma, x.mfma_addr, x.consumer, x.consumer_addr, x.have, x.need) for x in v[:20])


def test_schedule_variants_cover_every_matrix_instruction_source():
    """build.MFMA_SOURCES is derived from the sources, so a new MFMA translation unit is rebuilt under every schedule
    variant (round 5 left linear_rows_h.hip / pfn_fused.hip out of a hand-kept list), and the probe exercises them."""
    import glob
    from mssvt_amd import build
    want = {os.path.basename(p) for p in glob.glob(os.path.join(build.CSRC, "*.hip")) if "mfma" in open(p).read()}
    assert want - set(build.MFMA_EXCLUDED) == set(build.MFMA_SOURCES)
    assert {"linear_rows_h.hip", "pfn_fused.hip", "ffn.hip", "block_attn.hip", "compress_ws.hip"} <= set(build.MFMA_SOURCES)
    assert set(build.MFMA_EXCLUDED) == {"ceiling.hip"}  # timing-only launches: no output depends on them
    probe = open(os.path.join(os.path.dirname(build.HERE), "tools", "schedule_probe.py")).read()
    assert "vfe_fused" in probe and "linear_rows_h_%d_%d" in probe and "(256, 128)" in probe
