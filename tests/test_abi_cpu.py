"""CPU: the C-ABI library builds/loads and exports every symbol include/hfasr_hip.h declares, with the
argument counts the ctypes binding uses (no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "hfasr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(?<!\*\s)\b(int|size_t|void)\s+(mi_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a for a in m.group(3).split(",") if a.strip()]
        out[m.group(2)] = len(args)
    return out


@pytest.fixture(scope="module")
def built():
    from huggingface_asr_amd.csrc import build as B
    return B.build()


def test_header_symbols_exported(built):
    from huggingface_asr_amd import _lib
    h = _lib.lib()
    decl = _header_functions()
    assert len(decl) >= 16
    for name, nargs in decl.items():
        assert hasattr(h, name), f"{name} declared in hfasr_hip.h but not exported"
        if name.startswith("mi_profile_"):
            continue
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(_lib.SIGNATURES[name]) == nargs, f"{name}: header has {nargs} args, binding {len(_lib.SIGNATURES[name])}"
    assert set(_lib.SIGNATURES) == {n for n in decl if not n.startswith("mi_profile_")}
    assert hasattr(h, "mi_last_error")


def test_workspace_query_runs_on_cpu(built):
    import ctypes as C

    from huggingface_asr_amd import _lib
    from huggingface_asr_amd.engine import EBranchformerEngine
    from huggingface_asr_amd import shapes
    eng = EBranchformerEngine(dict(shapes.BASE), device="cpu")
    cs = eng._config_struct(32, 1000, 80)
    n = _lib.lib().mi_ebf_workspace_bytes(C.byref(cs))
    assert 400e6 < n < 1.5e9
    assert eng.out_frames(1000) == 250 and eng.out_frames(998) == 250


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from huggingface_asr_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryError):
        _lib.lib()


def test_library_code_objects_hold_no_packed_f32_arithmetic():
    """DESIGN.md 'Concurrent kernels': the library must stay free of v_pk_{add,mul,fma}_f32; build.py checks it after linking, this re-checks the objects
    of the build that produced the .so under test (skipped when only the .so travelled, e.g. on the GPU box)."""
    import glob
    import os
    import pytest
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(os.path.dirname(here), "huggingface_asr_amd", "csrc")
    objs = sorted(glob.glob(os.path.join(csrc, "build", "*.o")))
    if not objs or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no object files / no llvm-objdump here")
    sys.path.insert(0, csrc)
    import build
    assert len(objs) == len(build.SOURCES)
    build.check_no_packed_f32(objs)
