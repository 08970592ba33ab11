"""CPU: the C-ABI library loads and exports every symbol include/meatmodeler.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "meatmodeler.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from meatmodeler_amd import _lib
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/meatmodeler.h but not exported"
    # and the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_sizes_without_gpu():
    from meatmodeler_amd import _lib
    assert _lib.lib.mm_abi_version() == 3
    assert _lib.lib.mm_bf_workspace_bytes(1, 4000, 4000) > 0
    assert _lib.lib.mm_chol_workspace_bytes(3000) >= 47 * 64 * 64 * 8
    p = _lib.OrbParams(4000, 8, 31, 20, 1.2, 0)
    assert _lib.lib.mm_orb_workspace_bytes(2, 1080, 1920, ctypes.byref(p)) > 2 * 1920 * 1080
    assert ctypes.sizeof(_lib.BAProblem) == 16 + 8 * 8 + 16 + 2 * 8 + 8 + 6 * 8   # matches sizeof(mm_ba_problem)


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product path must refuse to run, not silently compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from meatmodeler_amd import _lib
    with pytest.raises(_lib.MMError):
        _lib.Context()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "meatmodeler_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
