"""The drop-in boundary on CPU: the C-ABI library builds for gfx950, loads, and
exports every symbol include/gar_hip.h declares.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gar_hip.h")


@pytest.fixture(scope="module")
def lib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "aligator_amd", "csrc")], check=True)
    from aligator_amd import _lib
    return _lib.load()


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gar_hip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(lib):
    from aligator_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gar_hip.h but not exported"
    assert set(declared) == set(_lib.SIGNATURES), "ctypes table out of sync with gar_hip.h"


def test_library_is_a_gfx950_code_object(lib):
    so = os.path.join(ROOT, "aligator_amd", "libgar_hip.so")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", so], capture_output=True, text=True).stdout
    assert ".hip_fatbin" in out
    strs = subprocess.run(["strings", so], capture_output=True, text=True).stdout
    assert "gfx950" in strs and "gar_backward_generic" in strs


def test_version_and_layout_queries(lib):
    assert b"gfx950" in lib.gar_hip_version()
    d = (C.c_int32 * 5)(36, 12, 0, 36, 0)
    assert lib.gar_hip_knot_doubles(d) == 3684
    assert lib.gar_hip_factor_doubles(d) == 3108     # ff 48 + fb 1728 + Vxx 1296 + vx 36
    d = (C.c_int32 * 5)(36, 12, 32, 36, 0)
    assert lib.gar_hip_knot_doubles(d) == 3684 + 32 * 36 + 32 * 12 + 32


def test_suggested_leg_count(lib):
    """The measured table behind gar_hip_suggest_num_legs (profiles/r06_seam_leg_counts.log): pure host logic."""
    assert lib.gar_hip_suggest_num_legs(256, 36, 12) == 64
    assert lib.gar_hip_suggest_num_legs(256, 56, 22) == 32
    assert lib.gar_hip_suggest_num_legs(2048, 36, 12) == 512
    assert lib.gar_hip_suggest_num_legs(5, 12, 4) == 2
    assert lib.gar_hip_suggest_num_legs(1, 12, 4) == 2 and lib.gar_hip_suggest_num_legs(0, 12, 4) == 1


def test_no_gpu_fails_loudly(lib):
    """There is no CPU fallback: without a HIP device solver creation raises."""
    if lib.gar_hip_device_count() > 0:
        pytest.skip("a HIP device is visible")
    from aligator_amd import synth
    from aligator_amd.gar import ProximalRiccatiSolver
    prob = synth.generate_lq_problem(1, np.zeros(2), 3, 2, 2)
    with pytest.raises(RuntimeError, match="no such HIP device"):
        ProximalRiccatiSolver(prob)


def test_missing_library_fails_loudly(tmp_path):
    from aligator_amd import _lib
    with pytest.raises(_lib.GarLibraryError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "libgar_hip.so"))


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under aligator_amd/ or include/ may
    import, link or execute it."""
    for base in ("aligator_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                    txt = open(os.path.join(dp, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle|gar_oracle|libgar_oracle", txt, re.M), f
