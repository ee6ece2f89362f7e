"""Importable name of the product package directory `vln-goat_amd/` (a hyphen is not a valid Python identifier): this package's
search path IS that directory, so `vln_goat_amd.hipops`, `vln_goat_amd.rollout`, ... are the modules there; the three public
names of its `__init__` are re-exported by plain imports."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'vln-goat_amd')]

from .layers import compute_dtype, set_compute_dtype  # noqa: E402,F401
from .hipops import manual_seed  # noqa: E402,F401

__version__ = '0.1.0'
