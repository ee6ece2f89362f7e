"""C-ABI checks that need no GPU: the library loads, exports every symbol include/goat_hip.h declares,
the ctypes signature table covers exactly that set, and argument validation returns error codes (no kernel
is launched: null pointers / bad shapes are rejected before any launch)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'goat_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\bint\s+(goat_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_are_exported_and_bound():
    from vln_goat_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    names = _declared()
    assert len(names) >= 15
    h = _lib.lib()
    for n in names:
        assert hasattr(h, n), 'libgoat_hip.so does not export %s' % n
    assert sorted(_lib.SIGNATURES.keys()) == names


def test_argument_validation_without_gpu():
    from vln_goat_amd import _lib
    h = _lib.lib()
    assert h.goat_version() >= 100
    assert h.goat_gemm_nt(None, 1, 1, None, 0, None, 0, None, 0, 4, 4, 8, None, 0, None, 0, 1) == -1      # GOAT_E_ARG
    assert h.goat_gemm_bf16(None, 0, 0, 1, None, 0, None, 0, None, 0, 4, 4, 64, None, 0, None, 0, 1, 128, 2, None) == -1
    buf = (ctypes.c_char * 256)()
    p = ctypes.addressof(buf) & ~15
    assert h.goat_gemm_bf16(None, 0, 0, 1, p, 8, p, 8, p, 8, 4, 4, 8, None, 0, None, 0, 1, 128, 2, None) == -2   # Kc % 64
    assert h.goat_ln_fwd(None, 1, None, None, None, None, 1e-5, 0.0, 0, 0, None, None, None, None, None, 4, 768) == -1
    assert h.goat_attn_fwd(None, 1, p, 64, 64, p, 64, 64, p, 64, 64, p, 64, 64, None, None, p, 1, 1, 4, 300, 0.125,
                           0.0, 0, 0, None) == -2                                                             # Lk > 256
    assert h.goat_ln_bwd_ws_floats(768) > 0


def test_product_ops_refuse_cpu_tensors():
    import torch
    from vln_goat_amd import hipops
    with pytest.raises(RuntimeError):
        hipops.linear(torch.zeros(4, 8), torch.nn.Parameter(torch.zeros(8, 8)), None)
