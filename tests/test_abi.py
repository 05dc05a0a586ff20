"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950,
loads, and exports every symbol include/rato_saa.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "rato_saa.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rato_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from riskaversetrajopt_amd import _build, _lib
    _build.build()
    return _lib.load()


def test_header_declares_expected_entry_points():
    fns = header_functions()
    for name in ("rato_drone_eval", "rato_drone_linearize", "rato_car_eval", "rato_car_linearize",
                 "rato_hopper_slip", "rato_sum_partials", "rato_risk_stats", "rato_scp_run_drone",
                 "rato_hopper_emit_jacobian_values", "rato_hopper_slip_hessian"):
        assert name in fns


def test_library_exports_every_declared_symbol(lib):
    from riskaversetrajopt_amd import _lib
    raw = ctypes.CDLL(_lib.lib_path())
    for name in header_functions():
        assert hasattr(raw, name), f"{name} declared in rato_saa.h but not exported"
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_abi_version_and_size_queries(lib):
    from riskaversetrajopt_amd import _lib
    assert lib.rato_abi_version() == _lib.ABI_VERSION == 12
    import ctypes as C
    cpt, spl, tile = C.c_int32(8), C.c_int32(1), C.c_int32(0)
    assert lib.rato_drone_linearize_plan(1000, 50, 1000, C.byref(cpt), C.byref(spl), C.byref(tile)) == 4
    assert tile.value == 256
    cpt, spl = C.c_int32(0), C.c_int32(0)
    nblk = lib.rato_drone_linearize_plan(100000, 50, 100000, C.byref(cpt), C.byref(spl), C.byref(tile))
    assert cpt.value == -1 and tile.value == 64 and nblk == (100000 + 63) // 64       # row-parallel default
    cpt, spl = C.c_int32(0), C.c_int32(0)
    assert lib.rato_drone_linearize_plan(1000, 500, 1000, C.byref(cpt), C.byref(spl), C.byref(tile)) > 0
    assert cpt.value > 0 and tile.value == 256                                          # LDS tables too big
    cpt, spl = C.c_int32(4), C.c_int32(4)
    assert lib.rato_drone_linearize_plan(1001, 50, 1001, C.byref(cpt), C.byref(spl), C.byref(tile)) < 0   # ld % 4
    assert lib.rato_drone_linearize_plan(0, 50, 0, C.byref(cpt), C.byref(spl), C.byref(tile)) < 0
    S = 40
    assert lib.rato_car_ego_scratch_floats(S) == (S + 1) * 4 + (S + 1) * 2 + (S + 1) * 2 * 2 * S
    assert lib.rato_risk_stats_workspace_bytes(10000) > 5120 * 4


def test_params_struct_layout_matches_header():
    from riskaversetrajopt_amd import _lib
    # 2 int32 + 6 float + 6 + 6 + 6 floats
    assert ctypes.sizeof(_lib.DroneParams) == 4 * (3 + 6 + 18 + 1) + 8 * (6 + 6 + 6 + 6) + 32   # + the fp64 constants + stats_*
    assert ctypes.sizeof(_lib.CarParams) == 4 * (2 + 5 + 8 + 1) + 8 * (4 + 4) + 32   # + the fp64 constants + stats_*


def test_cut_loop_struct_layouts_match_the_library(lib):
    from riskaversetrajopt_amd import _lib
    assert ctypes.sizeof(_lib.CutConfig) == lib.rato_cut_config_bytes()
    assert ctypes.sizeof(_lib.CutResult) == lib.rato_cut_result_bytes()
    assert ctypes.sizeof(_lib.ScpIter) == lib.rato_scp_iter_bytes()          # the per-iteration record of rato_scp_run_drone


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from riskaversetrajopt_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib._build, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RatoError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "riskaversetrajopt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
