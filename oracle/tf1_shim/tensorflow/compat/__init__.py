from . import v1  # noqa: F401
