"""stand-in package, see oracle/tf1_shim/README.md"""
from . import compat  # noqa: F401

__version__ = "0.0-shim"
_is_cornac_oracle_shim = True
