"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/cornac_hip.h declares.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from cornac_amd import _lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "cornac_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cornac_hip_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built_and_loads():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = _lib.lib()
    assert b"gfx950" in L.cornac_hip_version()


def test_every_declared_symbol_is_exported():
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "missing export: " + name
    assert sorted(_lib.SYMBOLS) == declared, "cornac_amd/_lib.py binds a different symbol set than the header declares"


def test_code_object_targets_gfx950_only():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_error_reporting_without_a_device():
    """Bad arguments come back as status codes + message, never as exceptions/aborts."""
    L = _lib.lib()
    import numpy as np

    h = ctypes.c_void_p()
    indptr = np.array([0, 1], np.int32)
    indices = np.array([0], np.int32)
    rc = L.cornac_hip_bpr_create(ctypes.byref(h), 0, 1, 1, 1, 1, 0, indptr, indices, 1)  # k = 0
    assert rc == 1 and b"k must be positive" in L.cornac_hip_last_error()
    if _lib.device_count() == 0:
        rc = L.cornac_hip_bpr_create(ctypes.byref(h), 0, 1, 1, 1, 1, 4, indptr, indices, 1)
        assert rc == 3, "without a GPU the library must fail loudly (no CPU fallback)"
        assert b"no CPU fallback" in L.cornac_hip_last_error()
        with pytest.raises(_lib.HipError):
            _lib.BprTrainer(indptr, indices, 1, 1, 1, 1, 4)


def test_product_never_imports_the_oracle_or_the_test_doubles():
    """the oracle is the checker: nothing under cornac_amd/ (nor bench.py outside its cpu_baseline leg) may route through it"""
    import ast

    pkg = os.path.join(ROOT, "cornac_amd")
    for name in sorted(os.listdir(pkg)):
        if not name.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, name)).read())
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                mods = [node.module or ""]
            for m in mods:
                assert not m.split(".")[0] in ("oracle", "fake_device", "tests"), (name, m)
    bench = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    users = {fn.name for fn in ast.walk(bench) if isinstance(fn, ast.FunctionDef)
             for node in ast.walk(fn) if isinstance(node, ast.ImportFrom) and (node.module or "").startswith("oracle")}
    assert users <= {"cpu_baseline", "cpu_baseline_reference", "cpu_rank_baseline", "cpu_baseline_mf", "cpu_baseline_vbpr", "leg_wmf_netflix"}, users
