"""cornac_amd — MI355X (gfx950) backend for the embedding-SGD + scoring hot path of PreferredAI/cornac.

Public surface mirrors the reference for this path: `BPR`, `WBPR`, `VEBPR`, `MF`, `VBPR`, `WMF` (models with the
reference's `Recommender.fit/score/rank/recommend/save/load/clone` interface), `Dataset`, `Reader`, and the callers `RatioSplit` / `BaseMethod` / `Experiment` (+ `eval`, `metrics`).
All compute runs in libcornac_hip.so (hand-written HIP for gfx950, C ABI in include/cornac_hip.h);
there is no CPU fallback.
"""
from .data import Dataset, FeatureModality, ImageModality, PurchaseViewDataset
from .reader import Reader
from .recommender import Recommender, ScoreException, adopt_reference_classes
from .bpr import BPR, WBPR, VEBPR
from .mf import MF
from .vbpr import VBPR
from .wmf import WMF
from .experiment import BaseMethod, CrossValidation, CVResult, Experiment, RatioSplit, Result, StratifiedSplit
from . import eval, metrics  # noqa: A004,F401

__all__ = ["Dataset", "PurchaseViewDataset", "Reader", "FeatureModality", "ImageModality", "RatioSplit", "StratifiedSplit", "CrossValidation", "CVResult", "BaseMethod", "Experiment", "Result", "Recommender", "ScoreException", "adopt_reference_classes", "BPR", "WBPR", "VEBPR", "MF", "VBPR", "WMF"]
__version__ = "0.1.0"
