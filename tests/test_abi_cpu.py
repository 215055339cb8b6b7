"""The C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol that
include/mssvt_hip.h declares (no compute calls here: there is no GPU in this container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mssvt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mssvt_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_declared_abi():
    from mssvt_amd import build
    path = build.build()
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libmssvt_hip.so does not export %s" % n
    lib.mssvt_hip_status_string.restype = ctypes.c_char_p
    assert lib.mssvt_hip_abi_version() >= 100
    assert lib.mssvt_hip_status_string(0) == b"ok"
    assert b"bad argument" in lib.mssvt_hip_status_string(-1)
    lib.mssvt_hash_workspace_ints.restype = ctypes.c_longlong
    assert lib.mssvt_hash_workspace_ints(1000, 2) >= 1000


def test_every_declared_entry_point_gets_its_argument_types():
    """mssvt_amd._lib declares argtypes / restype of every entry point from the header (the fused path then hands plain
    ints and addresses to ctypes): every declared symbol must be covered, with one ctypes type per parameter."""
    from mssvt_amd import _lib
    lib = _lib.lib()
    assert _lib.TYPED
    src = open(os.path.join(ROOT, "include", "mssvt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name in declared_symbols():
        fn = getattr(lib, name)
        assert fn.argtypes is not None, name
        m = re.search(r"\b%s\s*\(([^;{]*?)\)\s*;" % name, src, flags=re.S)
        params = m.group(1).strip()
        want = 0 if params in ("", "void") else params.count(",") + 1
        assert len(fn.argtypes) == want, (name, len(fn.argtypes), want)
    assert lib.mssvt_ffn_packed_bytes(128, 256) == 2 * 2 * 128 * 256 * 2  # plain ints in, long long out
    assert lib.mssvt_hip_status_string(0) == b"ok"


def test_argument_errors_are_status_codes_not_exits():
    from mssvt_amd import build
    lib = ctypes.CDLL(build.build())
    null = ctypes.c_void_p(0)
    i = ctypes.c_int
    # null pointers / bad sizes are rejected before any HIP call
    assert lib.mssvt_build_mapping_with_hash(i(8), i(8), i(8), i(10), i(0), i(1), null, null, null, null, null) == -1
    assert lib.mssvt_group_features(i(0), i(1), i(1), i(1), null, null, null, null, null, null) == -1
    assert lib.mssvt_three_nn(i(1), i(0), i(1), null, null, null, null, null) == -1


def test_product_has_no_oracle_import_and_fails_loudly_without_library(monkeypatch):
    import mssvt_amd
    pkg = os.path.dirname(mssvt_amd.__file__)
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            txt = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, fn
    from mssvt_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmssvt_hip.so")
    try:
        _lib.lib()
    except _lib.MssvtHipError as e:
        assert "no CPU fallback" in str(e).replace("\n", " ") or "not built" in str(e)
    else:
        raise AssertionError("missing library must raise")


def test_pybind_stand_ins_cover_every_call_of_the_reference_python():
    """Every `mssvt_ops_cuda.<f>(...)` / `pointnet2.<f>(...)` call in the reference's own op files that the path uses
    must exist in the stand-in modules with the same number of positional parameters (build container only: the
    reference tree is not shipped)."""
    import ast
    import inspect
    ref = "/root/reference/pcdet/ops"
    files = {"mssvt_ops_cuda": (os.path.join(ref, "mssvt", "mssvt_ops.py"), "mssvt_amd.mssvt_ops_compat"),
             "pointnet2": (os.path.join(ref, "pointnet2", "pointnet2_batch", "pointnet2_utils.py"),
                           "mssvt_amd.pointnet2_compat")}
    if not all(os.path.exists(f) for f, _ in files.values()):
        pytest.skip("reference tree not present")
    import importlib
    off_path = {"ball_query_wrapper", "three_interpolate_wrapper", "three_interpolate_grad_wrapper"}
    seen = 0
    for alias, (path, modname) in files.items():
        mod = importlib.import_module(modname)
        for node in ast.walk(ast.parse(open(path).read())):
            if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute)
                    and isinstance(node.func.value, ast.Name) and node.func.value.id == alias):
                name = node.func.attr
                assert hasattr(mod, name), "%s.%s missing" % (modname, name)
                if name in off_path:
                    continue
                params = inspect.signature(getattr(mod, name)).parameters
                assert len(params) == len(node.args), "%s: %d parameters, the reference passes %d" % (
                    name, len(params), len(node.args))
                seen += 1
    assert seen >= 10
