"""Build libmssvt_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m mssvt_amd.build [--force]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build
container.  The .so is git-ignored but travels to the GPU box with the snapshot.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmssvt_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: every fused multiply-add in the library is written explicitly
# (fmaf / MFMA), so results do not depend on the compiler's contraction choices.
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent scalar fp32 FMAs into v_pk_fma_f32
# and pays for it with register shuffles (v_mov): 2x the VALU instructions in the
# LDS-fed mat-vec loops of block_attn.hip (measured from the ISA).
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


# Schedule variants of the translation units that issue matrix instructions (tests/test_mfma_schedules_gpu.py): the same
# sources under another optimisation level / without the post-RA scheduler.  A result that is "right by the luck of the
# schedule" (a sum read before its last matrix instruction has landed) differs between them; every
# variant must reproduce the default library's outputs.  Only the listed files are recompiled, the rest is re-linked.
# The list is DERIVED (every csrc/*.hip whose text names a matrix instruction), so a new MFMA translation unit cannot be
# left out; ceiling.hip holds timing-only launches that no output depends on.
MFMA_EXCLUDED = ("ceiling.hip",)


def mfma_sources():
    out = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        base = os.path.basename(src)
        if base in MFMA_EXCLUDED:
            continue
        with open(src) as f:
            if "mfma" in f.read():
                out.append(base)
    return tuple(out)


MFMA_SOURCES = mfma_sources()
VARIANTS = {"O2": ["-O2"], "nopost": ["-mllvm", "-enable-post-misched=false"]}
VARIANT_DIR = os.path.join(LIB_DIR, "variants")


def variant_path(tag):
    return os.path.join(VARIANT_DIR, "libmssvt_hip_%s.so" % tag)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False):
    """Compile every HIP source into one shared library; returns its path."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    # objects (and compiler temporaries) whose source is gone must not linger beside the library
    keep = {os.path.basename(s) + ".o" for s in sources()} | {os.path.basename(LIB_PATH)}
    for name in os.listdir(LIB_DIR):
        if name not in keep and (".hip.o" in name) and os.path.isfile(os.path.join(LIB_DIR, name)):
            os.remove(os.path.join(LIB_DIR, name))
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIB_DIR, os.path.basename(src) + ".o")
        cmd = [HIPCC] + FLAGS + os.environ.get("MSSVT_EXTRA_HIPCC_FLAGS", "").split() + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed for %s:\n%s\n" % (src, out.decode()))
        elif verbose and out:
            print(out.decode())
    if failed:
        raise RuntimeError("hipcc compilation failed")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


def build_variants(force=False):
    """The schedule-variant libraries {tag: path} (built after `build()`, whose objects the other sources reuse)."""
    build()
    os.makedirs(VARIANT_DIR, exist_ok=True)
    out = {}
    for tag, extra in VARIANTS.items():
        path = variant_path(tag)
        out[tag] = path
        if not force and os.path.exists(path) and os.path.getmtime(path) >= os.path.getmtime(LIB_PATH):
            continue
        procs, objs = [], []
        for src in sources():
            base = os.path.basename(src)
            if base not in MFMA_SOURCES:
                objs.append(os.path.join(LIB_DIR, base + ".o"))
                continue
            obj = os.path.join(VARIANT_DIR, "%s.%s.o" % (base, tag))
            objs.append(obj)
            procs.append((src, subprocess.Popen([HIPCC] + FLAGS + extra + ["-c", src, "-o", obj], stdout=subprocess.PIPE,
                                                stderr=subprocess.STDOUT)))
        for src, p in procs:
            o, _ = p.communicate()
            if p.returncode != 0:
                raise RuntimeError("hipcc failed for %s (%s):\n%s" % (src, tag, o.decode()))
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", path] + objs)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--variants" in sys.argv:
        print(build_variants(force="--force" in sys.argv))
