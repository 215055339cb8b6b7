#!/usr/bin/env python3
"""Static MFMA-result hazard check over the gfx950 code objects of libmssvt_hip.so.

Why: gfx9 matrix instructions are NOT interlocked against the next reader of their destination registers -- the
compiler's hazard recogniser has to put enough independent instructions / `s_nop`s between a `v_mfma_*` and the first
instruction that reads or overwrites its result.  Round 4 met a build of `k_cmp_ws` whose sums were read without their
last term where control flow joined between the last MFMA of a chain and the first read: values
off by 1e-4, deterministically per build, invisible to every test until the schedule happened to change.  This tool
makes that class of error a BUILD failure: it disassembles the shipped library (`llvm-objdump -d` on every gfx950 code
object of the `.hip_fatbin` section), rebuilds each kernel's control-flow graph from the branch targets, and for every
`v_mfma_*` walks EVERY path forward until the required number of wait states has passed, flagging any instruction on the
way that touches the destination registers too early.

Wait states (one per issued instruction, `s_nop N` = N + 1; the model of LLVM's GCNHazardRecognizer): with P = passes of
the producing MFMA (4 cycles each),

  producer                          | VALU read / write, memory / LDS / export read | MFMA reads it as A or B | MFMA reads an overlapping, not identical, C
  XDL  (16-bit / 8-bit inputs)      | P + 3 (+1 on gfx950 when P != 2)              | as for VALU (1)         | P + 1 (+1 on gfx950 when P != 2)
  SGEMM (`*_f32` inputs, non-XDL)   | P + 2                                         | P + 2                   | P
  any, consumer MFMA takes the WHOLE destination as its C (accumulation chain)     : 0
  (1) the compiler pads 7 or 8 there depending on how the consumer's destination overlaps: the larger is required here

`calibrate()` compiles one-MFMA kernels with the same hipcc and reads the padding the compiler itself inserts in
straight-line code; tests/test_mfma_hazard_cpu.py asserts that the table above equals it, so a toolchain that changes the
rule is noticed as well.

    python tools/mfma_hazard_check.py [path/to/libmssvt_hip.so]      # exit code 1 on a violation
"""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# passes (4 clocks each) of the matrix instructions this library issues, and whether the matrix pipe treats them as XDL
# (16-bit inputs) or as SGEMM (f32 inputs).  A mnemonic that is not listed fails the check: add it with its passes.
MFMA = {
    "v_mfma_f32_16x16x4_f32": (8, False),
    "v_mfma_f32_32x32x2_f32": (16, False),
    "v_mfma_f32_4x4x1_16b_f32": (2, False),
    "v_mfma_f32_16x16x32_f16": (4, True),
    "v_mfma_f32_16x16x32_bf16": (4, True),
    "v_mfma_f32_16x16x16_f16": (4, True),
    "v_mfma_f32_16x16x16_bf16": (4, True),
    "v_mfma_f32_32x32x16_f16": (8, True),
    "v_mfma_f32_32x32x16_bf16": (8, True),
    "v_mfma_f32_32x32x8_f16": (8, True),
    "v_mfma_f32_32x32x8_bf16": (8, True),
}

VALU, MEM, MFMA_AB, MFMA_C = "valu", "mem", "mfma_ab", "mfma_c"


def required(mnemonic, kind, gfx950=True):
    """Wait states between `mnemonic` and a consumer of `kind` touching its destination."""
    passes, xdl = MFMA[mnemonic]
    bump = 1 if (gfx950 and passes != 2) else 0
    if xdl:
        return {VALU: passes + 3 + bump, MEM: passes + 3 + bump, MFMA_AB: passes + 3 + bump, MFMA_C: passes + 1 + bump}[kind]
    return {VALU: passes + 2, MEM: passes + 2, MFMA_AB: passes + 2, MFMA_C: passes}[kind]


# ------------------------------------------------------------------------------------------------------------------
# code objects -> instruction lists
# ------------------------------------------------------------------------------------------------------------------
def code_objects(lib_path):
    """The gfx950 ELF images inside the library's .hip_fatbin section (one clang offload bundle per translation unit)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM_BIN, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib_path, fat])
        data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        (n,) = struct.unpack_from("<Q", data, i + len(magic))
        o = i + len(magic) + 8
        for _ in range(n):
            off, size, tsz = struct.unpack_from("<QQQ", data, o)
            o += 24
            triple = data[o:o + tsz].decode()
            o += tsz
            if "gfx950" in triple and size:
                out.append(data[i + off:i + off + size])
        pos = i + len(magic)
    return out


Instr = collections.namedtuple("Instr", "addr mnem ops target")
_LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<([^>+]+)(?:\+0x([0-9a-fA-F]+))?>)?\s*$")
_FUNC = re.compile(r"^([0-9a-fA-F]+) <([^>]+)>:\s*$")


def parse_disassembly(text):
    """{function name: [Instr]} from `llvm-objdump -d` output (branch targets resolved to addresses)."""
    funcs, cur, start = collections.OrderedDict(), None, {}
    for line in text.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = m.group(2)
            start[cur] = int(m.group(1), 16)
            funcs[cur] = []
            continue
        if cur is None:
            continue
        m = _LINE.match(line)
        if not m:
            continue
        mnem, ops, addr, sym, off = m.group(1), m.group(2), int(m.group(3), 16), m.group(4), m.group(5)
        target = None
        if mnem.startswith("s_cbranch") or mnem == "s_branch":
            if sym is not None and sym in start:
                target = start[sym] + (int(off, 16) if off else 0)
            else:  # no symbolised target: relative word offset from the next instruction
                target = addr + 4 + 4 * _simm16(ops)
        funcs[cur].append(Instr(addr, mnem, ops, target))
    return funcs


def _simm16(ops):
    v = int(ops.strip().split()[0], 0)
    return v - 65536 if v >= 32768 else v


def disassemble(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".elf") as f:
        f.write(elf_bytes)
        f.flush()
        return subprocess.check_output([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", f.name]).decode()


# ------------------------------------------------------------------------------------------------------------------
# registers
# ------------------------------------------------------------------------------------------------------------------
_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs_of(operand):
    s = set()
    for m in _REG.finditer(operand):
        if m.group(2) is not None:
            s.add((m.group(1), int(m.group(2))))
        else:
            s.update((m.group(1), r) for r in range(int(m.group(3)), int(m.group(4)) + 1))
    return s


def split_operands(ops):
    return [o.strip() for o in ops.split(",")] if ops else []


_LOADS = ("global_load", "buffer_load", "flat_load", "scratch_load", "ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle",
          "ds_consume", "ds_append", "image_load", "image_sample", "s_")


def touched(ins):
    """(registers the instruction reads or overwrites in a hazard-relevant way, consumer kind or None)."""
    m = ins.mnem
    ops = split_operands(ins.ops)
    if m.startswith("v_mfma") or m.startswith("v_smfma"):
        return None, None  # handled by the caller (operand roles matter)
    if m.startswith("v_"):
        return set().union(*[regs_of(o) for o in ops]) if ops else set(), VALU
    if m.startswith(("global_", "buffer_", "flat_", "scratch_", "ds_", "exp", "image_", "tbuffer_")):
        is_load = m.startswith(_LOADS) or "_atomic" in m and False
        rd = ops[1:] if (is_load and " lds" not in ins.ops and not m.endswith("_lds")) else ops
        # returning atomics / loads write their first operand: a write by the memory pipe is not a hazard the
        # recogniser pads (the data returns long after the matrix result has landed); every other operand is READ
        return set().union(*[regs_of(o) for o in rd]) if rd else set(), MEM
    return set(), None


def wait_states(ins):
    if ins.mnem == "s_nop":
        return int(ins.ops.strip().split()[0], 0) + 1
    return 1


# ------------------------------------------------------------------------------------------------------------------
# the check
# ------------------------------------------------------------------------------------------------------------------
Violation = collections.namedtuple("Violation", "func mfma_addr mfma consumer_addr consumer kind have need")


def successors(instrs, index_of, i):
    ins = instrs[i]
    m = ins.mnem
    if m in ("s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_trap", "s_rfe_b64"):
        return []
    nxt = [i + 1] if i + 1 < len(instrs) else []
    if m == "s_branch":
        return [index_of[ins.target]] if ins.target in index_of else []
    if m.startswith("s_cbranch"):
        return nxt + ([index_of[ins.target]] if ins.target in index_of else [])
    return nxt


def check_function(name, instrs, strict=False):
    index_of = {ins.addr: i for i, ins in enumerate(instrs)}
    out = []
    for i, ins in enumerate(instrs):
        if not ins.mnem.startswith("v_mfma"):
            continue
        if ins.mnem not in MFMA:
            out.append(Violation(name, ins.addr, ins.mnem, ins.addr, "(unknown matrix instruction: add it to MFMA)", "table", 0, 0))
            continue
        ops = split_operands(ins.ops)
        dst = regs_of(ops[0])
        horizon = max(required(ins.mnem, k) for k in (VALU, MEM, MFMA_AB, MFMA_C))
        seen = {}
        # `free`: when the SIMD's matrix pipe takes the next matrix instruction (it is in order and busy for the passes of
        # the one before: measured issue intervals, /opt/skills/guides/MI355X_MICROARCH.md "cycle constants")
        stack = [(s, 0, frozenset(dst), MFMA[ins.mnem][0]) for s in successors(instrs, index_of, i)]
        while stack:
            # `live`: the registers that still hold THIS instruction's result on the path walked
            j, waited, live, free = stack.pop()
            if waited >= horizon or not live or seen.get((j, live), 1 << 30) <= waited:
                continue
            seen[(j, live)] = waited
            c = instrs[j]
            if c.mnem.startswith("v_mfma") and not strict and c.mnem in MFMA:
                waited = max(waited, free)  # it cannot ISSUE before the pipe is free
                free = waited + MFMA[c.mnem][0]
                if waited >= horizon:
                    continue
            if c.mnem.startswith("v_mfma"):
                cops = split_operands(c.ops)
                cdst, a, b, cc = regs_of(cops[0]), regs_of(cops[1]), regs_of(cops[2]), regs_of(cops[3])
                if (a | b) & live:
                    need = required(ins.mnem, MFMA_AB)
                    if waited < need:
                        out.append(Violation(name, ins.addr, ins.mnem, c.addr, c.mnem + " " + c.ops, MFMA_AB, waited, need))
                # C = the whole, untouched destination: an accumulation chain, ordered by the pipe (0 wait states)
                if cc & live and not (cc == dst and live == dst):
                    need = required(ins.mnem, MFMA_C)
                    if waited < need:
                        out.append(Violation(name, ins.addr, ins.mnem, c.addr, c.mnem + " " + c.ops, MFMA_C, waited, need))
                # (a later matrix instruction that OVERWRITES registers is ordered behind this one by the pipe itself;
                # from here on they hold its result, and its own walk covers their readers)
                live = live - cdst
            else:
                regs, kind = touched(c)
                if kind is not None and regs & live:
                    need = required(ins.mnem, kind)
                    if waited < need:
                        out.append(Violation(name, ins.addr, ins.mnem, c.addr, c.mnem + " " + c.ops, kind, waited, need))
                if kind == VALU:
                    ops_c = split_operands(c.ops)
                    if ops_c and not c.mnem.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                        live = live - regs_of(ops_c[0])  # overwritten (checked above as a write-after-write)
            w = waited + wait_states(c)
            for s in successors(instrs, index_of, j):
                stack.append((s, w, live, free))
    return out


def check_text(disassembly, strict=False):
    funcs = parse_disassembly(disassembly)
    out, n_mfma = [], 0
    for name, instrs in funcs.items():
        n_mfma += sum(1 for x in instrs if x.mnem.startswith("v_mfma"))
        out.extend(check_function(name, instrs, strict))
    return out, len(funcs), n_mfma


def check_library(lib_path, strict=False):
    """(violations, kernels, matrix instructions) over every gfx950 code object of the library."""
    out, nf, nm = [], 0, 0
    for elf in code_objects(lib_path):
        v, f, m = check_text(disassemble(elf), strict)
        out.extend(v)
        nf += f
        nm += m
    return out, nf, nm


# ------------------------------------------------------------------------------------------------------------------
# calibration against the compiler's own padding
# ------------------------------------------------------------------------------------------------------------------
_CAL_SRC = r"""
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
#define CAL(NAME, T, CALL)                                                        \
    extern "C" __global__ void cal_valu_##NAME(const T *a, const T *b, float *o) {           \
        f4 acc = {0, 0, 0, 0};                                                    \
        acc = CALL(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);                 \
        o[threadIdx.x] = acc[0] + acc[1];                                         \
    }                                                                             \
    extern "C" __global__ void cal_mem_##NAME(const T *a, const T *b, f4 *o) {               \
        f4 acc = {0, 0, 0, 0};                                                    \
        acc = CALL(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);                 \
        o[threadIdx.x] = acc;                                                     \
    }                                                                             \
    extern "C" __global__ void cal_ab_##NAME(const T *a, const T *b, const float *c, f4 *o) {\
        const float cv = c[threadIdx.x]; /* waited for before the first product */ \
        f4 acc = {cv, cv, cv, cv};                                                \
        acc = CALL(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);                 \
        f4 acc2 = {0, 0, 0, 0};                                                   \
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv, acc[1], acc2, 0, 0, 0);   \
        o[threadIdx.x] = acc2;                                                    \
    }
CAL(f32, float, __builtin_amdgcn_mfma_f32_16x16x4f32)
CAL(f16, h8, __builtin_amdgcn_mfma_f32_16x16x32_f16)
CAL(bf16, b8, __builtin_amdgcn_mfma_f32_16x16x32_bf16)
CAL(f16k16, h4, __builtin_amdgcn_mfma_f32_16x16x16f16)
"""
CAL_MNEMONIC = {"f32": "v_mfma_f32_16x16x4_f32", "f16": "v_mfma_f32_16x16x32_f16", "bf16": "v_mfma_f32_16x16x32_bf16",
                "f16k16": "v_mfma_f32_16x16x16_f16"}


def calibrate():
    """{(mnemonic, kind): wait states the compiler leaves between the instruction and its first consumer in straight-line
    code} for kind in VALU / MEM / MFMA_AB, from one-MFMA kernels compiled with the library's compiler."""
    with tempfile.TemporaryDirectory() as tmp:
        src, asm = os.path.join(tmp, "cal.hip"), os.path.join(tmp, "cal.s")
        open(src, "w").write(_CAL_SRC)
        subprocess.check_call([HIPCC, "-O3", "--offload-arch=gfx950", "--cuda-device-only", "-S", src, "-o", asm],
                              stderr=subprocess.DEVNULL)
        text = open(asm).read()
    out = {}
    for m in re.finditer(r"^(cal_(valu|mem|ab)_(\w+)):[^\n]*$(.*?)s_endpgm", text, re.S | re.M):
        kind, tag, body = {"valu": VALU, "mem": MEM, "ab": MFMA_AB}[m.group(2)], m.group(3), m.group(4)
        lines = [ln.split(";")[0].strip() for ln in body.splitlines()]
        lines = [ln for ln in lines if ln and not ln.startswith(".") and not ln.endswith(":")]
        first = next(i for i, ln in enumerate(lines) if ln.startswith("v_mfma"))
        dst = regs_of(lines[first].split(None, 1)[1].split(",")[0])
        waited = 0
        for ln in lines[first + 1:]:
            mnem, _, ops = ln.partition(" ")
            ins = Instr(0, mnem, ops.strip(), None)
            if mnem.startswith("v_mfma"):
                cops = split_operands(ins.ops)
                hit = (regs_of(cops[1]) | regs_of(cops[2])) & dst
            else:
                regs, k = touched(ins)
                hit = k is not None and regs & dst
            if hit:
                out[(CAL_MNEMONIC[tag], kind)] = waited
                break
            waited += wait_states(ins)
    return out


def main(argv):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    strict = "--strict" in argv
    argv = [a for a in argv if a != "--strict"]
    lib = argv[1] if len(argv) > 1 else os.path.join(here, "mssvt_amd", "lib", "libmssvt_hip.so")
    v, nf, nm = check_library(lib, strict)
    print("%s: %d kernels, %d matrix instructions, %d hazard violations (%s)" %
          (lib, nf, nm, len(v), "one wait state per instruction" if strict else "matrix pipe occupancy counted"))
    for x in v[:50]:
        print("  %s: %s @%x -> %s @%x (%s): %d wait states, %d required" %
              (x.func, x.mfma, x.mfma_addr, x.consumer, x.consumer_addr, x.kind, x.have, x.need))
    return 1 if v else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
