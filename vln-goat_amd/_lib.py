"""ctypes binding of libgoat_hip.so (the C ABI declared in include/goat_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
`torch` is imported first on purpose: libgoat_hip.so needs libamdhip64.so.7, and the dynamic loader then
re-uses the copy PyTorch already loaded, so HIP streams and device pointers are shared with torch.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('GOAT_HIP_LIB') or os.path.join(CSRC, 'libgoat_hip.so')     # (override: kernel A/B experiments)
SOURCES = ['gemm.hip', 'gemm2.hip', 'gemm3.hip', 'gemm5.hip', 'attention.hip', 'attention2.hip', 'rowops.hip', 'causal.hip', 'optim.hip', 'glue.hip']

GOAT_F32, GOAT_BF16 = 0, 1
EPI_NONE, EPI_GELU, EPI_RELU, EPI_MUL_DGELU, EPI_MUL_DRELU, EPI_ACCUM = 0, 1, 2, 3, 4, 5

_lib = None

_vp = ctypes.c_void_p
_i32 = ctypes.c_int
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64
_f32 = ctypes.c_float

# name -> argtypes; must list every symbol include/goat_hip.h declares (checked by tests/test_abi.py)
SIGNATURES = {
    'goat_version': [],
    'goat_gemm_nt': [_vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _i64, _i32],
    'goat_gemm_bf16': [_vp, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp, _i64, _i32, _i32, _i32, _vp],
    'goat_colsum': [_vp, _i32, _vp, _i64, _i32, _i32, _vp],
    'goat_transpose': [_vp, _i32, _vp, _i64, _vp, _i64, _i32, _i32, _vp],
    'goat_ln_fwd': [_vp, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _i32, _i32],
    'goat_wgrad_smallk': [_vp, _i32, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp],
    'goat_ln_bwd_ws_floats': [_i32],
    'goat_ln_bwd_nparts': [_i32],
    'goat_ln_reduce_batched': [_vp, _vp, _i32, _i32],
    'goat_ln_bwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    'goat_ln_fwd_do': [_vp, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _u64, _vp],
    'goat_ln_bwd_do': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _f32, _u64, _vp],
    'goat_dropout_add_fwd': [_vp, _i32, _vp, _vp, _vp, _i64, _f32, _u64, _u64, _vp],
    'goat_dropout_bwd': [_vp, _i32, _vp, _vp, _i64, _f32, _u64, _u64, _vp],
    'goat_act_bwd': [_vp, _i32, _vp, _vp, _vp, _i64, _i32, _f32, _u64, _u64, _vp],
    'goat_attn_fwd': [_vp, _i32] + [_vp, _i64, _i64] * 4 + [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _u64, _u64, _vp],
    'goat_attn_bwd': [_vp, _i32] + [_vp, _i64, _i64] * 8 + [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _u64, _u64, _vp],
    'goat_ce_fwd': [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp],
    'goat_ce_bwd': [_vp, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64],
    'goat_pano_fusion_fwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32],
    'goat_pano_fusion_bwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32],
    'goat_gather_segmean_fwd': [_vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp],
    'goat_gather_segmean_bwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32],
    'goat_embed_fwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp],
    'goat_embed_bwd': [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32],
    'goat_attn_pool_fwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    'goat_attn_pool_bwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32],
    'goat_door_gate_fwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32],
    'goat_door_gate_bwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp],
    'goat_dict_wsum_fwd': [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32],
    'goat_dict_wsum_bwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32],
    'goat_wgrad_grouped': [_vp, _vp, _i32, _i32, _i32],
    'goat_wgrad_grouped_balanced': [_vp, _vp, _i32, _i32, _vp, _i64],
    'goat_wgrad_balanced_ws_bytes': [_i32],
    'goat_grad_sqnorm': [_vp, _vp, _vp, _i32, _vp],
    'goat_adamw_step': [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _f32, _f32, _f32, _vp],
    'goat_sap_fuse_fwd': [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32],
    'goat_sap_fuse_bwd': [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32],
    'goat_infonce_fwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32],
    'goat_infonce_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32],
    'goat_probe_tr16': [_vp, _vp],
    'goat_add_n': [_vp, _i32, _vp, _i32, _vp, _i64],
    'goat_rowdot_fwd': [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32],
    'goat_rowdot_bwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32],
    'goat_cfp_mix_fwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32],
    'goat_cfp_mix_bwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32],
    'goat_zero_ranges': [_vp, _vp, _vp, _i32],
}


class LnPartial(ctypes.Structure):
    """struct goat_ln_partial (include/goat_hip.h)."""
    _fields_ = [('ws', ctypes.c_void_p), ('dgamma', ctypes.c_void_p), ('dbeta', ctypes.c_void_p), ('nparts', ctypes.c_int32),
                ('reserved', ctypes.c_int32)]


class AdamwTensor(ctypes.Structure):
    """struct goat_adamw_tensor (include/goat_hip.h)."""
    _fields_ = [('param', ctypes.c_void_p), ('shadow0', ctypes.c_void_p), ('shadow1', ctypes.c_void_p), ('arena_off', ctypes.c_int64),
                ('numel', ctypes.c_int64), ('step_size', ctypes.c_float), ('decay', ctypes.c_float), ('shadow_f32', ctypes.c_void_p),
                ('cols', ctypes.c_int32), ('ld0', ctypes.c_int32)]


class WgradProblem(ctypes.Structure):
    """struct goat_wgrad_problem (include/goat_hip.h)."""
    _fields_ = [('dy', ctypes.c_void_p), ('ld_dy', ctypes.c_int64), ('x', ctypes.c_void_p), ('ld_x', ctypes.c_int64),
                ('dw', ctypes.c_void_p), ('ld_dw', ctypes.c_int64), ('dbias', ctypes.c_void_p),
                ('rows', ctypes.c_int), ('n_out', ctypes.c_int), ('n_in', ctypes.c_int), ('accumulate', ctypes.c_int)]


LAST_BUILD = [None]        # what the last build() call did: 'reused ...' or 'compiled ...' (__graft_entry__.build prints it)


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into csrc/libgoat_hip.so (in-tree, travels with the repo snapshot)."""
    force = force or os.environ.get('GOAT_FORCE_BUILD', '0') == '1'
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, 'common.hpp'), os.path.join(CSRC, 'gemm2_tile.hpp'), os.path.join(CSRC, 'gemm5_tile.hpp'), os.path.join(CSRC, 'attn_args.hpp'), os.path.join(_HERE, '..', 'include', 'goat_hip.h')]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        LAST_BUILD[0] = 'reused (library newer than every source; GOAT_FORCE_BUILD=1 or build(force=True) recompiles)'
        return LIB_PATH
    LAST_BUILD[0] = 'compiled %d sources with hipcc --offload-arch=gfx950' % len(srcs)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + '.o'
        objs.append(o)
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), out.decode()))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('link failed: %s\n%s' % (' '.join(cmd), r.stdout.decode()))
    return LIB_PATH


def lib():
    """Load (once) and return the ctypes handle; raises loudly if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libgoat_hip.so not found at %s — the GOAT HIP kernels are required (no CPU/eager fallback). '
                'Run `python -c "import __graft_entry__ as g; g.build()"` from the repo root.' % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, argt in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = argt
            fn.restype = ctypes.c_int
        _lib = h
    return _lib


def check(status, what):
    if status != 0:
        raise RuntimeError('libgoat_hip: %s failed with status %d' % (what, status))
