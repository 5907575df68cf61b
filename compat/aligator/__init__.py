"""Opt-in shim: `import aligator.gar` served by the MI355X backend (aligator_amd), for scripts written against
the reference's Python bindings (bindings/python/src/gar/expose-*.cpp).  ONLY the `gar` submodule exists here --
nothing else of aligator is rebuilt.  Put this directory on PYTHONPATH *instead of* an
aligator installation, never beside one:

    PYTHONPATH=/path/to/repo:/path/to/repo/compat python my_lqr_script.py
"""
from . import gar  # noqa: F401

__all__ = ["gar"]
