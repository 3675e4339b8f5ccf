"""TEST INFRASTRUCTURE ONLY — import the real reference hot path (this container only).

`load()` makes `cornac.models.bpr`, `cornac.models.mf`, `cornac.data`,
`cornac.utils`, `cornac.metrics` importable from /root/reference WITHOUT running
`cornac/__init__.py` / `cornac/models/__init__.py` (which eagerly import all 66
models and 24 compiled extensions).  The six hot-path extensions come from
oracle/_ref/ (built by oracle/build_ref.py); everything else is the reference's
own unmodified Python.

Nothing under cornac_amd/ may import this module; only tests/ and
tests/golden/make_golden.py do.  /root/reference does not exist on the GPU box,
so `available()` is False there and callers must skip.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

from . import build_ref

REF = build_ref.REF
_loaded = False


def available():
    return build_ref.available() and all(os.path.exists(build_ref.so_path(r)) for r in build_ref.EXTS)


class _RefExtFinder(importlib.abc.MetaPathFinder):
    """Resolve the hot-path extension modules to the .so files in oracle/_ref."""

    def __init__(self):
        self.table = {r.replace("/", "."): build_ref.so_path(r) for r in build_ref.EXTS}

    def find_spec(self, fullname, path, target=None):
        so = self.table.get(fullname)
        if so is None:
            return None
        loader = importlib.machinery.ExtensionFileLoader(fullname, so)
        return importlib.machinery.ModuleSpec(fullname, loader, origin=so)


def _stub_pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def load():
    """Returns a namespace with the reference classes of the hot path."""
    global _loaded
    if not available():
        raise RuntimeError("reference not available (no /root/reference or oracle/_ref not built)")
    if not _loaded:
        if "cornac" in sys.modules and not getattr(sys.modules["cornac"], "_oracle_stub", False):
            raise RuntimeError("a real `cornac` is already imported")
        sys.meta_path.insert(0, _RefExtFinder())
        root = _stub_pkg("cornac", os.path.join(REF, "cornac"))
        root._oracle_stub = True
        _stub_pkg("cornac.models", os.path.join(REF, "cornac", "models"))
        # eval_methods/__init__ pulls `powerlaw` (absent, unused on this path)
        sys.modules.setdefault("powerlaw", types.ModuleType("powerlaw"))
        _loaded = True
    import importlib

    ns = types.SimpleNamespace()
    ns.recom_bpr = importlib.import_module("cornac.models.bpr.recom_bpr")
    ns.recom_wbpr = importlib.import_module("cornac.models.bpr.recom_wbpr")
    ns.BPR = ns.recom_bpr.BPR
    ns.RNGVector = ns.recom_bpr.RNGVector
    ns.WBPR = ns.recom_wbpr.WBPR
    ns.MF = importlib.import_module("cornac.models.mf").MF
    ns.backend_cpu = importlib.import_module("cornac.models.mf.backend_cpu")
    ns.fast_dot = importlib.import_module("cornac.utils.fast_dot").fast_dot
    ns.Dataset = importlib.import_module("cornac.data").Dataset
    rec = importlib.import_module("cornac.models.recommender")
    ns.Recommender = rec.Recommender
    for cls in ("Recommender", "NextBasketRecommender", "NextItemRecommender"):
        setattr(sys.modules["cornac.models"], cls, getattr(rec, cls))
    ns.metrics = importlib.import_module("cornac.metrics")
    ns.eval_methods = importlib.import_module("cornac.eval_methods")
    return ns


def kernel_available():
    """the compiled reference kernel (oracle/_ref) can be loaded, with or without /root/reference"""
    need = ("cornac/models/bpr/recom_bpr", "cornac/utils/fast_dot")
    return all(os.path.exists(build_ref.so_path(r)) for r in need)


def load_kernel_only():
    """RNGVector and BPR of the REAL compiled reference extension, loaded over the stand-in Python modules of
    oracle/ref_stubs when /root/reference is absent (the GPU box): enough to drive and time `BPR._fit_sgd`
    (bench.py cpu_baseline kind "reference").  Where the reference tree exists, load() is used instead."""
    global _loaded
    if available():
        ns = load()
        return ns.RNGVector, ns.BPR
    if not kernel_available():
        raise RuntimeError("oracle/_ref is not built")
    import importlib

    if not _loaded:
        if "cornac" in sys.modules:
            raise RuntimeError("a `cornac` package is already imported")
        stubs = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_stubs")
        sys.meta_path.insert(0, _RefExtFinder())
        sys.path.insert(0, stubs)
        _loaded = True
    m = importlib.import_module("cornac.models.bpr.recom_bpr")
    return m.RNGVector, m.BPR


def load_mf_kernel():
    """`fit_sgd` of the REAL compiled reference extension cornac/models/mf/backend_cpu (oracle/_ref).  The extension
    only imports numpy / multiprocessing / tqdm, so on the GPU box it loads over an empty stand-in package."""
    global _loaded
    import importlib

    if available():
        return load().backend_cpu.fit_sgd
    if not os.path.exists(build_ref.so_path("cornac/models/mf/backend_cpu")):
        raise RuntimeError("oracle/_ref is not built")
    if not _loaded:
        if "cornac" in sys.modules:
            raise RuntimeError("a `cornac` package is already imported")
        stubs = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_stubs")
        sys.meta_path.insert(0, _RefExtFinder())
        sys.path.insert(0, stubs)
        _loaded = True
    return importlib.import_module("cornac.models.mf.backend_cpu").fit_sgd
