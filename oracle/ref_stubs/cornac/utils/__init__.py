"""stub package: see ../../README.md (fast_dot is the reference's own compiled extension from oracle/_ref)"""
from .common import get_rng  # noqa: F401
from .fast_dot import fast_dot  # noqa: F401  (resolved to oracle/_ref by the loader's finder)
