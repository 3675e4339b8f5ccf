"""TEST INFRASTRUCTURE ONLY — the reference's own WMF (cornac/models/wmf/recom_wmf.py + wmf.py, unmodified) in this
container.  TensorFlow is absent from the image: when `import tensorflow` fails, oracle/tf1_shim (a stand-in for the
symbols those two files use, torch underneath) is put on sys.path first.  See oracle/tf1_shim/README.md for what the
stand-in computes and what it restates."""
import importlib
import os
import sys

from . import ref_loader


def available():
    return ref_loader.available()


def uses_shim():
    import tensorflow

    return bool(getattr(tensorflow, "_is_cornac_oracle_shim", False))


def load_wmf():
    ref_loader.load()
    try:
        importlib.import_module("tensorflow")
    except ImportError:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf1_shim"))
        importlib.import_module("tensorflow")
    return importlib.import_module("cornac.models.wmf").WMF
