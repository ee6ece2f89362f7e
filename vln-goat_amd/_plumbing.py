"""Pointer / dtype / stream helpers shared by the modules that call the C ABI (hipops, tuning, wgrad_queue)."""
import torch

from ._lib import GOAT_BF16, GOAT_F32


def _dt(t):
    if t.dtype == torch.float32:
        return GOAT_F32
    if t.dtype == torch.bfloat16:
        return GOAT_BF16
    raise RuntimeError('libgoat_hip supports float32 / bfloat16, got %s' % t.dtype)


def _epc(t):
    return 4 if t.dtype == torch.float32 else 8


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError('GOAT HIP ops need tensors on the GPU (no CPU fallback in the product path)')


def _ptr(t, off=0):
    return t.data_ptr() + off * t.element_size()
