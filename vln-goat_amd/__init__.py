"""vln-goat_amd — MI355X-native (gfx950) implementation of GOAT's cross-modal transformer hot path.

Import as `vln_goat_amd` (the sibling alias package maps the importable name onto this directory).
Contents: csrc/ (HIP kernels + C ABI), _lib.py (ctypes binding), hipops.py (autograd ops),
layers.py / pretrain_model.py (reference-compatible nn.Module tree), graphmap.py (host index building),
dp.py (data-parallel engine over RCCL), synth.py (synthetic batches), config.py.
"""
from .layers import compute_dtype, set_compute_dtype  # noqa: F401
from .hipops import manual_seed  # noqa: F401

__version__ = '0.1.0'
