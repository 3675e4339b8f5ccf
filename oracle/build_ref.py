#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — builds the *real* reference hot path into oracle/_ref/.

The reference (PreferredAI/cornac, mounted read-only at /root/reference) is a
Python package whose hot loops are Cython.  This recipe cythonises and compiles
ONLY the extensions on the hot path, straight from the sources where they lie
under /root/reference, with the reference's own compiler flags
(`setup.py:128-138`: -O3 -ffast-math -fopenmp -std=c++11, vendored boost 1.72):

    cornac/models/bpr/recom_bpr.pyx    (BPR._fit_sgd, RNGVector, has_non_zero)
    cornac/models/bpr/recom_wbpr.pyx   (WBPR.fit)
    cornac/models/bpr/recom_vebpr.pyx  (imported by cornac/models/bpr/__init__.py)
    cornac/models/mf/backend_cpu.pyx   (MF fit_sgd)
    cornac/utils/fast_dot.pyx          (score() mat-vec)
    cornac/utils/fast_sparse_funcs.pyx (imported by cornac/utils/common.py)

Outputs (generated .cpp and the .so files) go ONLY under oracle/_ref/ (git-ignored).
No reference source is copied into this repository.  The resulting extensions
import the reference's *Python* modules at load time, so they are usable only
where /root/reference exists (this container) — see oracle/ref_loader.py.  They
are used to (1) validate the C restatement in oracle/*.c and (2) generate the
golden vectors committed under tests/golden/ (tests/golden/make_golden.py).

This does NOT run the reference's own build system (setup.py).
"""
import os
import subprocess
import sys
import sysconfig

REF = os.environ.get("CORNAC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

EXTS = [
    "cornac/utils/fast_sparse_funcs",
    "cornac/utils/fast_dot",
    "cornac/models/bpr/recom_bpr",
    "cornac/models/bpr/recom_wbpr",
    "cornac/models/bpr/recom_vebpr",
    "cornac/models/mf/backend_cpu",
]


def available():
    return os.path.isdir(os.path.join(REF, "cornac"))


def ext_suffix():
    return sysconfig.get_config_var("EXT_SUFFIX")


def so_path(rel):
    return os.path.join(OUT, rel + ext_suffix())


def build(force=False, verbose=True):
    if not available():
        if verbose:
            print("[build_ref] %s not present: skipping (GPU box uses golden fixtures)" % REF)
        return False
    import numpy as np

    gen_dir = os.path.join(OUT, "gen")
    os.makedirs(gen_dir, exist_ok=True)
    py_inc = sysconfig.get_paths()["include"]
    np_inc = np.get_include()
    boost_inc = os.path.join(REF, "cornac", "utils", "external")
    cflags = ["-O3", "-ffast-math", "-Wno-unused-function", "-Wno-maybe-uninitialized",
              "-std=c++11", "-fopenmp", "-fPIC", "-shared", "-w",
              "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION"]
    procs = []
    for rel in EXTS:
        pyx = os.path.join(REF, rel + ".pyx")
        so = so_path(rel)
        if (not force) and os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(pyx):
            continue
        os.makedirs(os.path.dirname(so), exist_ok=True)
        cpp = os.path.join(gen_dir, rel.replace("/", "__") + ".cpp")
        cmd = [sys.executable, "-m", "cython", "--cplus", "-3", "-I", REF, "-o", cpp, pyx]
        if verbose:
            print("[build_ref] cython", rel)
        subprocess.check_call(cmd, cwd=REF)
        gxx = ["g++"] + cflags + ["-I", py_inc, "-I", np_inc, "-I", boost_inc,
                                  "-I", os.path.join(REF, os.path.dirname(rel)), cpp, "-o", so]
        procs.append((rel, subprocess.Popen(gxx)))
    for rel, p in procs:
        if p.wait() != 0:
            raise RuntimeError("g++ failed for " + rel)
        if verbose:
            print("[build_ref] built", so_path(rel))
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv)
    sys.exit(0 if ok or not available() else 1)
