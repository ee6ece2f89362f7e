"""vln-goat_amd — MI355X-native (gfx950) implementation of GOAT's cross-modal transformer hot path.

Import as `vln_goat_amd` (the sibling alias package maps the importable name onto this directory).
Contents: csrc/ (HIP kernels + C ABI), _lib.py (ctypes binding), hipops.py (autograd ops, GEMM autotuner, deferred grouped
weight gradients), layers.py / pretrain_model.py / nav_model.py (reference-compatible nn.Module trees for pre-training
and navigation fine-tuning), graphmap.py (host index building), dp.py (gradient arena + data-parallel engine over RCCL),
synth.py (synthetic batches / episodes), config.py, tuned_gfx950.json (autotuned GEMM table).
"""
from .layers import compute_dtype, set_compute_dtype  # noqa: F401
from .hipops import manual_seed  # noqa: F401

__version__ = '0.1.0'
