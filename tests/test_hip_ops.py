"""GPU parity tests of the individual HIP kernels (through the C ABI) against plain PyTorch fp32
references of the same op.  Tolerances: fp32 path 1e-3 relative-to-scale (north_star), bf16 path 2e-2."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _tol(dtype):
    return 1e-3 if dtype == torch.float32 else 2e-2


def _close(got, ref, dtype, what=''):
    got, ref = got.float().cpu(), ref.float().cpu()
    scale = ref.abs().max().clamp_min(1e-6)
    err = (got - ref).abs().max() / scale
    assert err < _tol(dtype), '%s: max err / scale = %.3e' % (what, err)


@pytest.fixture(scope='module')
def ops():
    from vln_goat_amd import hipops
    return hipops


def test_library_loads_and_version():
    from vln_goat_amd import _lib
    assert _lib.lib().goat_version() >= 100


def test_probe_tr16_semantics(ops):
    """ds_read_b64_tr_b16 lane mapping (recorded for the TN-GEMM work; asserts the documented hypothesis:
    within a 16-lane group, result(l, j) = source lane (4*j + (l&15)>>2), element (l&3))."""
    from vln_goat_amd import _lib
    out = torch.zeros(256, dtype=torch.int16, device=DEV)
    st = _lib.lib().goat_probe_tr16(torch.cuda.current_stream().cuda_stream, out.data_ptr())
    assert st == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.int64).reshape(64, 4)
    import os
    os.makedirs('gpurun_out', exist_ok=True)
    np.savetxt('gpurun_out/tr16_probe.txt', got, fmt='%d')
    exp = np.zeros((64, 4), dtype=np.int64)
    for l in range(64):
        g, i = l // 16, l % 16
        for j in range(4):
            src_lane = g * 16 + 4 * j + (i >> 2)
            exp[l, j] = src_lane * 4 + (i & 3)
    assert np.array_equal(got, exp), got[:16]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 768, 768), (37, 1, 768), (300, 2304, 768), (65, 50, 3072),
                                   (1056, 768, 8)])
def test_gemm_nt_plain_and_bias(ops, dtype, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(DEV, dtype)
    b = (torch.randn(N, K, generator=g) * 0.1).to(DEV, dtype)
    bias = torch.randn(N, generator=g).to(DEV)
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    ops.gemm_nt(a, b, out, bias)
    ref = a.float() @ b.float().T + bias
    _close(out, ref, dtype, 'gemm')
    # asymmetric operands catch transposes: also check a single known element
    i, j = M // 2, N // 3
    assert abs(float(out[i, j]) - float(ref[i, j])) <= _tol(dtype) * float(ref.abs().max())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('act', ['gelu', 'relu'])
def test_gemm_epilogues(ops, dtype, act):
    from vln_goat_amd._lib import EPI_GELU, EPI_RELU, EPI_MUL_DGELU, EPI_MUL_DRELU
    M, N, K = 257, 3072, 768
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).to(DEV, dtype)
    b = (torch.randn(N, K, generator=g) * 0.05).to(DEV, dtype)
    bias = torch.randn(N, generator=g).to(DEV) * 0.1
    out = torch.empty(M, N, device=DEV, dtype=dtype)
    aux = torch.empty(M, N, device=DEV, dtype=dtype)
    ops.gemm_nt(a, b, out, bias, EPI_GELU if act == 'gelu' else EPI_RELU, aux)
    u = a.float() @ b.float().T + bias
    ref = torch.nn.functional.gelu(u) if act == 'gelu' else torch.relu(u)
    _close(aux, u, dtype, 'pre-activation')
    _close(out, ref, dtype, 'activation')
    # backward epilogue: C = (A·B^T) * act'(aux)
    dy = torch.randn(M, K, generator=g).to(DEV, dtype)     # reuse shapes: dX[M,N'] with N' = N, K' = K
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV, dtype)
    dxo = torch.empty(M, N, device=DEV, dtype=dtype)
    ops.gemm_nt(dy, w, dxo, None, EPI_MUL_DGELU if act == 'gelu' else EPI_MUL_DRELU, aux)
    uu = aux.float().requires_grad_(True)
    (torch.nn.functional.gelu(uu) if act == 'gelu' else torch.relu(uu)).sum().backward()
    ref2 = (dy.float() @ w.float().T) * uu.grad
    _close(dxo, ref2, dtype, 'act-derivative epilogue')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_wgrad_splitk_and_bias_grad(ops, dtype):
    M, N, K = 1003, 768, 3072
    g = torch.Generator().manual_seed(11)
    dy = (torch.randn(M, N, generator=g) * 0.1).to(DEV, dtype)
    x = torch.randn(M, K, generator=g).to(DEV, dtype)
    dw, db = ops.wgrad(dy, x, True)
    _close(dw, dy.float().T @ x.float(), dtype, 'dW')
    _close(db, dy.float().sum(0), dtype, 'db')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('R,C', [(64, 64), (100, 37), (3840, 768), (5, 3072)])
def test_transpose_pad(ops, dtype, R, C):
    x = torch.randn(R, C).to(DEV, dtype)
    cs = torch.zeros(C, device=DEV)
    out = ops.transpose_pad(x, cs)
    assert out.shape[0] == C and out.shape[1] >= R
    assert torch.equal(out[:, :R], x.T)
    assert float(out[:, R:].abs().sum()) == 0.0
    _close(cs, x.float().sum(0), dtype, 'colsum')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('with_res', [False, True])
def test_layer_norm_fwd_bwd(ops, dtype, with_res):
    M, H = 515, 768
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, H, generator=g).to(DEV, dtype).requires_grad_(True)
    r = torch.randn(M, H, generator=g).to(DEV, dtype).requires_grad_(True) if with_res else None
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.1 * torch.randn(H, generator=g)).to(DEV).requires_grad_(True)
    y = ops.layer_norm(x, gamma, beta, 1e-12, r, 0.0)
    dy = torch.randn(M, H, generator=g).to(DEV, dtype)
    y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    rr = r.detach().float().requires_grad_(True) if with_res else None
    gr, br = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    z = xr + rr if with_res else xr
    yr = torch.nn.functional.layer_norm(z, (H,), gr, br, 1e-12)
    yr.backward(dy.float())
    _close(y, yr, dtype, 'ln y')
    _close(x.grad, xr.grad, dtype, 'ln dx')
    if with_res:
        _close(r.grad, rr.grad, dtype, 'ln dres')
    _close(gamma.grad, gr.grad, dtype, 'ln dgamma')
    _close(beta.grad, br.grad, dtype, 'ln dbeta')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('K', [7, 14])
@pytest.mark.parametrize('sunk', [False, True])
def test_linear_short_input_weight_gradient(ops, dtype, K, sunk):
    """Linear(7 | 14 -> 768) (the position-feature Linears): forward through the K-padded GEMM, weight / bias gradients by
    goat_wgrad_smallk — returned to autograd, or added straight into bound gradient slices (as under the gradient arena)."""
    rows, N = 1003, 768
    g = torch.Generator().manual_seed(K)
    x = torch.randn(rows, K, generator=g).to(DEV, dtype)
    w = (0.2 * torch.randn(N, K, generator=g)).to(DEV).requires_grad_(True)
    b = (0.1 * torch.randn(N, generator=g)).to(DEV).requires_grad_(True)
    dy = torch.randn(rows, N, generator=g).to(DEV, dtype)
    base_w, base_b = torch.randn(N, K, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    if sunk:
        for prm, snk in ((w, base_w.clone()), (b, base_b.clone())):
            prm.grad = snk
            prm.__dict__['_goat_sink'] = snk
            prm.__dict__['_goat_prezero'] = True          # "cleared at step start": the kernel only ever adds
    try:
        y = ops.linear(x, w, b)
        y.backward(dy)
        torch.cuda.synchronize()
    finally:
        for prm in (w, b):
            prm.__dict__.pop('_goat_sink', None)
            prm.__dict__.pop('_goat_prezero', None)
    wr, br = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.linear(x.float(), wr, br)
    yr.backward(dy.float())
    _close(y, yr, dtype, 'short linear y')
    _close(w.grad - (base_w if sunk else 0), wr.grad, dtype, 'short linear dW')
    _close(b.grad - (base_b if sunk else 0), br.grad, dtype, 'short linear db')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_layer_norm_fork_in_sums_the_skip_gradient(ops, dtype):
    """layer_norm(fork_in=True) -> (LN(x), x): the gradient of the second output (the skip connection of a pre-LN block) is added
    to dx inside goat_ln_bwd (dx_add); also when only one of the two outputs carries a gradient."""
    M, H = 777, 768
    g = torch.Generator().manual_seed(17)
    x = torch.randn(M, H, generator=g).to(DEV, dtype).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.1 * torch.randn(H, generator=g)).to(DEV).requires_grad_(True)
    d1 = torch.randn(M, H, generator=g).to(DEV, dtype)
    d2 = torch.randn(M, H, generator=g).to(DEV, dtype)
    y, skip = ops.layer_norm(x, gamma, beta, 1e-5, fork_in=True)
    assert skip.data_ptr() == x.data_ptr()
    torch.autograd.backward([y, skip], [d1, d2])
    xr = x.detach().float().requires_grad_(True)
    gr, br = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-5)
    torch.autograd.backward([yr, xr * 1.0], [d1.float(), d2.float()])
    _close(y, yr, dtype, 'ln fork_in y')
    _close(x.grad, xr.grad, dtype, 'ln fork_in dx + skip')
    _close(gamma.grad, gr.grad, dtype, 'ln fork_in dgamma')
    _close(beta.grad, br.grad, dtype, 'ln fork_in dbeta')
    x.grad = None
    y, skip = ops.layer_norm(x, gamma, beta, 1e-5, fork_in=True)
    skip.backward(d2)                                  # the LayerNorm branch unused
    assert torch.equal(x.grad, d2)
    x.grad = None
    y, skip = ops.layer_norm(x, gamma, beta, 1e-5, fork_in=True)
    y.backward(d1)                                     # the skip branch unused
    xr.grad = None
    torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-5).backward(d1.float())
    _close(x.grad, xr.grad, dtype, 'ln fork_in dx only')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_layer_norm_deferred_column_reduction(ops, dtype):
    """goat_ln_bwd(accumulate = 2) + goat_ln_reduce_batched: two LayerNorm calls that share gamma / beta (bound to
    pre-zeroed gradient slices, as under the gradient arena) and a third with its own leave their column partials behind;
    the one batched reduction at the end of the backward pass adds them up — same values as torch autograd."""
    H = 768
    g = torch.Generator().manual_seed(31)
    mk = lambda *s: torch.randn(*s, generator=g)
    params = [(1 + 0.1 * mk(H)).to(DEV).requires_grad_(True) for _ in range(2)] + [(0.1 * mk(H)).to(DEV).requires_grad_(True) for _ in range(2)]
    g1, g2, b1, b2 = params
    sinks = [torch.zeros(H, device=DEV) for _ in params]
    for prm, snk in zip(params, sinks):
        prm.grad = snk
        prm.__dict__['_goat_sink'] = snk
        prm.__dict__['_goat_prezero'] = True
    xs = [mk(m, H).to(DEV, dtype).requires_grad_(True) for m in (515, 3840, 1056)]
    dys = [mk(m, H).to(DEV, dtype) for m in (515, 3840, 1056)]
    try:
        assert ops.LnReduceQueue.enabled
        ys = [ops.layer_norm(xs[0], g1, b1, 1e-12), ops.layer_norm(xs[1], g1, b1, 1e-12), ops.layer_norm(xs[2], g2, b2, 1e-12)]
        torch.autograd.backward(ys, dys)
        assert not ops.LnReduceQueue.items            # flushed by the end-of-backward callback
        torch.cuda.synchronize()
    finally:
        for prm in params:
            prm.__dict__.pop('_goat_sink', None)
            prm.__dict__.pop('_goat_prezero', None)
    xr = [x.detach().float().requires_grad_(True) for x in xs]
    pr = [q.detach().clone().requires_grad_(True) for q in params]
    yr = [torch.nn.functional.layer_norm(xr[0], (H,), pr[0], pr[2], 1e-12), torch.nn.functional.layer_norm(xr[1], (H,), pr[0], pr[2], 1e-12),
          torch.nn.functional.layer_norm(xr[2], (H,), pr[1], pr[3], 1e-12)]
    torch.autograd.backward(yr, [d.float() for d in dys])
    for a, b in zip(xs, xr):
        _close(a.grad, b.grad, dtype, 'ln deferred dx')
    for snk, q, n in zip(sinks, pr, ['dgamma shared', 'dgamma own', 'dbeta shared', 'dbeta own']):
        _close(snk, q.grad, dtype, 'ln deferred ' + n)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dropout_mask_consistency_and_rate(ops, dtype):
    ops.manual_seed(123)
    n = 1 << 20
    x = torch.ones(n, device=DEV, dtype=dtype, requires_grad=True)
    y = ops.dropout(x, 0.1)
    keep = (y != 0)
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - 0.1) < 5e-3
    assert torch.allclose(y[keep].float(), torch.full_like(y[keep].float(), 1 / 0.9), rtol=1e-2)
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad != 0, keep)          # backward regenerates the same mask
    y2 = ops.dropout(x, 0.1)                        # a second call draws a different mask
    assert not torch.equal(y2 != 0, keep)
    # LayerNorm-fused dropout: residual path untouched, dropped path scaled
    ops.manual_seed(7)
    xx = torch.randn(256, 768, device=DEV).to(dtype).requires_grad_(True)
    rr = torch.randn(256, 768, device=DEV).to(dtype).requires_grad_(True)
    gmm = torch.ones(768, device=DEV, requires_grad=True)
    bt = torch.zeros(768, device=DEV, requires_grad=True)
    yy = ops.layer_norm(xx, gmm, bt, 1e-5, rr, 0.3)
    yy.backward(torch.randn_like(yy))
    frac_zero = (xx.grad == 0).float().mean().item()
    assert abs(frac_zero - 0.3) < 2e-2
    assert (rr.grad == 0).float().mean().item() < 1e-2


def _attn_ref(q, k, v, kmask, bias, nh):
    B, Lq, H = q.shape
    Lk = k.shape[1]

    def sp(x):
        return x.view(x.shape[0], x.shape[1], nh, 64).permute(0, 2, 1, 3)
    s = sp(q) @ sp(k).transpose(-1, -2) / 8.0
    if kmask is not None:
        s = s + kmask[:, None, None, :]
    if bias is not None:
        s = s + bias[:, None]
    p = torch.softmax(s, -1)
    return (p @ sp(v)).permute(0, 2, 1, 3).reshape(B, Lq, H)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('Lq,Lk,mode,use_bias,inf', [
    (80, 80, 'self', False, False), (36, 36, 'self', False, True), (22, 22, 'self', True, False),
    (37, 80, 'cross', False, False), (80, 23, 'cross', False, False), (130, 130, 'self', False, False),
    (5, 200, 'cross', False, False), (33, 65, 'cross', False, False),
    # 129..256-row sequences on the LDS-staged kernels: two query tiles per wave / several backward roles per wave
    (160, 160, 'self', False, False), (60, 200, 'cross', False, False), (200, 38, 'cross', False, False), (256, 256, 'self', False, True),
    (200, 200, 'self', True, False)])
def test_attention_fwd_bwd(ops, dtype, Lq, Lk, mode, use_bias, inf):
    B, nh, H = 3, 12, 768
    g = torch.Generator().manual_seed(Lq * 131 + Lk)
    klens = torch.tensor([Lk, max(1, Lk // 2), max(1, Lk - 3)])
    valid = torch.arange(Lk)[None, :] < klens[:, None]
    kmask = torch.zeros(B, Lk).masked_fill(~valid, float('-inf') if inf else -10000.0).to(DEV)
    bias = (torch.randn(B, Lq, Lk, generator=g) * 0.5).to(DEV).requires_grad_(True) if use_bias else None
    if mode == 'self':
        a = (torch.randn(B, Lq, 3 * H, generator=g) * 0.7).to(DEV, dtype).requires_grad_(True)
        b = None
        o = ops.attention(a, None, kmask, bias, nh, 0.0)
        af = a.detach().float().requires_grad_(True)
        q, k, v = af.split(H, -1)
    else:
        a = (torch.randn(B, Lq, H, generator=g) * 0.7).to(DEV, dtype).requires_grad_(True)
        b = (torch.randn(B, Lk, 2 * H, generator=g) * 0.7).to(DEV, dtype).requires_grad_(True)
        o = ops.attention(a, b, kmask, bias, nh, 0.0)
        af = a.detach().float().requires_grad_(True)
        bf = b.detach().float().requires_grad_(True)
        q = af
        k, v = bf.split(H, -1)
    biasf = bias.detach().clone().requires_grad_(True) if use_bias else None
    ref = _attn_ref(q, k, v, kmask, biasf, nh)
    do = torch.randn(B, Lq, H, generator=g).to(DEV)
    o.backward(do.to(dtype))
    ref.backward(do.to(dtype).float())
    _close(o, ref, dtype, 'attn out')
    _close(a.grad, af.grad, dtype, 'attn da')
    if b is not None:
        _close(b.grad, bf.grad, dtype, 'attn db')
    if use_bias:
        _close(bias.grad, biasf.grad, dtype, 'attn dbias')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('L', [64, 160, 200, 256])
def test_attention_dropout_backward_uses_the_forward_mask(ops, dtype, L):
    """The backward pass must regenerate the FORWARD's dropout mask whatever kernel family serves each direction (the LDS-staged
    backward needs more LDS than the staged forward: at L = 193..256 the forward runs staged and the backward streams).  The mask
    is recovered from forward calls on one-hot V (O[q, d] != 0 <=> probability (q, 64 j + d) kept), then a torch reference with
    that mask gives the expected gradients."""
    B, nh, H, p = 2, 12, 768, 0.3
    g = torch.Generator().manual_seed(L)
    qk = (torch.randn(B, L, 2 * H, generator=g) * 0.5).to(DEV, dtype)
    v = (torch.randn(B, L, H, generator=g) * 0.7).to(DEV, dtype)
    keep = torch.zeros(B, nh, L, L, dtype=torch.bool, device=DEV)
    for j in range((L + 63) // 64):
        onehot = torch.zeros(B, L, nh, 64, device=DEV, dtype=dtype)
        n = min(64, L - 64 * j)
        onehot[:, 64 * j + torch.arange(n), :, torch.arange(n)] = 1
        ops.manual_seed(1234)
        o = ops.attention(torch.cat([qk, onehot.view(B, L, H)], -1), None, None, None, nh, p)
        keep[:, :, :, 64 * j:64 * j + n] = (o.view(B, L, nh, 64).permute(0, 2, 1, 3)[..., :n] != 0)
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - p) < 0.02, rate
    a = torch.cat([qk, v], -1).requires_grad_(True)
    ops.manual_seed(1234)
    o = ops.attention(a, None, None, None, nh, p)
    do = torch.randn(B, L, H, generator=g).to(DEV, dtype)
    o.backward(do)
    af = a.detach().float().requires_grad_(True)

    def sp(x):
        return x.view(B, L, nh, 64).permute(0, 2, 1, 3)
    qf, kf, vf = af.split(H, -1)
    pr = torch.softmax(sp(qf) @ sp(kf).transpose(-1, -2) / 8.0, -1) * keep / (1.0 - p)
    ref = (pr @ sp(vf)).permute(0, 2, 1, 3).reshape(B, L, H)
    ref.backward(do.float())
    _close(o, ref, dtype, 'attn dropout out')
    _close(a.grad, af.grad, dtype, 'attn dropout grads')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_attention_dropout_is_unbiased_and_reproducible(ops, dtype):
    B, L, nh, H = 4, 64, 12, 768
    ops.manual_seed(99)
    a = (torch.randn(B, L, 3 * H) * 0.5).to(DEV, dtype).requires_grad_(True)
    base = ops.attention(a, None, None, None, nh, 0.0).float()
    acc = torch.zeros_like(base)
    n = 64
    for _ in range(n):
        acc += ops.attention(a, None, None, None, nh, 0.1).float()
    mean = acc / n
    assert (mean - base).abs().mean() / base.abs().mean() < 0.08
    # backward with dropout: finite-difference-free check via linearity in dO
    ops.manual_seed(5)
    o1 = ops.attention(a, None, None, None, nh, 0.1)
    o1.backward(torch.ones_like(o1))
    g1 = a.grad.clone()
    a.grad = None
    ops.manual_seed(5)
    o2 = ops.attention(a, None, None, None, nh, 0.1)
    assert torch.equal(o1, o2)
    o2.backward(torch.ones_like(o2))
    assert torch.equal(g1, a.grad)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_pano_fusion(ops, dtype):
    N, V, H = 17, 36, 768
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, V, H, generator=g).to(DEV, dtype).requires_grad_(True)
    w = (torch.randn(1, H, generator=g) * 0.05).to(DEV).requires_grad_(True)
    b0 = torch.randn(1, generator=g).to(DEV).requires_grad_(True)
    f = ops.pano_fusion(x, w, b0)
    df = torch.randn(N, H, generator=g).to(DEV)
    f.backward(df.to(dtype))
    xr = x.detach().float().requires_grad_(True)
    wr, br = w.detach().clone().requires_grad_(True), b0.detach().clone().requires_grad_(True)
    ws = torch.softmax(torch.tanh(xr @ wr.T + br), dim=1)
    fr = (xr * ws).sum(1)
    fr.backward(df.to(dtype).float())
    _close(f, fr, dtype, 'fused')
    _close(x.grad, xr.grad, dtype, 'dx')
    _close(w.grad, wr.grad, dtype, 'da')
    _close(b0.grad, br.grad, dtype, 'da0')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gather_segmean(ops, dtype):
    rows, H, n_out = 50, 768, 9
    g = torch.Generator().manual_seed(8)
    src = torch.randn(rows, H, generator=g).to(DEV, dtype).requires_grad_(True)
    segs = [[], [3], [4, 4, 7], [49, 0], [], [1, 2, 3, 5], [10], [11, 12], []]
    idx = torch.tensor([i for s in segs for i in s] or [-1], dtype=torch.int32, device=DEV)
    start = torch.tensor(np.cumsum([0] + [len(s) for s in segs]), dtype=torch.int32, device=DEV)
    scale = torch.tensor([1.0 / max(1, len(s)) for s in segs], device=DEV)
    out = ops.gather_segmean(src, idx, start, scale, n_out)
    dout = torch.randn(n_out, H, generator=g).to(DEV, dtype)
    out.backward(dout)
    sr = src.detach().float().requires_grad_(True)
    ref = torch.stack([sr[s].mean(0) if s else torch.zeros(H, device=DEV) for s in segs])
    ref.backward(dout.float())
    _close(out, ref, dtype, 'gather')
    _close(src.grad, sr.grad, dtype, 'gather grad')
    # with the inverse index the backward pass is a gather over dout (one writer per source row): same values, source dtype
    from vln_goat_amd import graphmap
    for sc in (scale, None):
        inv = tuple(t.to(DEV) for t in graphmap.inverse_index(idx, start, sc, rows) if t is not None)
        s2 = src.detach().clone().requires_grad_(True)
        o2 = ops.gather_segmean(s2, idx, start, sc, n_out, inv)
        o2.backward(dout)
        s3 = src.detach().clone().requires_grad_(True)
        ops.gather_segmean(s3, idx, start, sc, n_out).backward(dout)
        assert s2.grad.dtype == dtype and s2.grad.shape == src.shape
        _close(s2.grad, s3.grad, dtype, 'gather grad via inverse index')
        assert float(s2.grad[20:49].abs().max()) == 0.0          # rows nobody reads: exact zeros, written (no stale memory)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('with_types', [False, True])
def test_embedding_tables(ops, dtype, with_types):
    """word + type + position lookups in one kernel; scatter-add backward with nn.Embedding(padding_idx) semantics
    (the gradient of the padding row of the word table AND of position row `pad` is zero, as in the reference)."""
    import torch.nn.functional as F
    B, L, H, V, P, pad = 7, 19, 768, 300, 40, 1
    g = torch.Generator().manual_seed(33)
    word = torch.nn.Parameter(torch.randn(V, H, generator=g).to(DEV))
    typ = torch.nn.Parameter(torch.randn(2, H, generator=g).to(DEV))
    pos = torch.nn.Parameter(torch.randn(P, H, generator=g).to(DEV))
    ids = torch.randint(0, V, (B, L), generator=g).to(DEV)
    ids[:, -4:] = pad                          # padded tail
    ids[0, :5] = 17                            # repeated ids (atomics accumulate)
    tids = torch.randint(0, 2, (B, L), generator=g).to(DEV) if with_types else None
    out = ops.embedding(ids, word, typ, tids, pos, out_dtype=dtype, word_pad=pad, pos_pad=pad)
    dout = torch.randn(B, L, H, generator=g).to(DEV, dtype)
    out.backward(dout)
    ops.check_embed_errors()
    wr, tr, pr = [p.detach().clone().requires_grad_(True) for p in (word, typ, pos)]
    ref = F.embedding(ids, wr, padding_idx=pad) + F.embedding(tids if with_types else torch.zeros_like(ids), tr) \
        + F.embedding(torch.arange(L, device=DEV)[None], pr, padding_idx=pad)
    ref.backward(dout.float())
    _close(out, ref, dtype, 'embedding')
    for a, b, n in ((word, wr, 'word'), (typ, tr, 'type'), (pos, pr, 'pos')):
        assert a.grad.dtype == torch.float32
        _close(a.grad, b.grad, torch.float32, 'embedding grad ' + n)      # f32 accumulation of the same dout values
    assert float(word.grad[pad].abs().max()) == 0.0 and float(pos.grad[pad].abs().max()) == 0.0
    # single-table form (gmap_step_embeddings) + out-of-range detection
    step = torch.nn.Parameter(torch.randn(15, H, generator=g).to(DEV))
    sid = torch.randint(0, 15, (4, 22), generator=g).to(DEV)
    o2 = ops.embedding(sid, step, out_dtype=dtype)
    _close(o2, F.embedding(sid, step), dtype, 'embedding single')
    o2.backward(torch.ones_like(o2))
    _close(step.grad, torch.bincount(sid.reshape(-1), minlength=15).float()[:, None].expand(15, H), torch.float32, 'step grad')
    # many rows into a 3-row table (nav-type embeddings of a batch of panoramas): the LDS-accumulating backward kernel
    small = torch.nn.Parameter(torch.randn(3, H, generator=g).to(DEV))
    nid = torch.randint(0, 3, (240, 37), generator=g).to(DEV)
    o3 = ops.embedding(nid, small, out_dtype=dtype)
    d3 = torch.randn(240, 37, H, generator=g).to(DEV, dtype)
    o3.backward(d3)
    sr = small.detach().clone().requires_grad_(True)
    F.embedding(nid, sr).backward(d3.float())
    _close(small.grad, sr.grad, torch.float32, 'small-table grad')
    bad = sid.clone()
    bad[0, 0] = 15
    ops.embedding(bad, step, out_dtype=dtype)
    with pytest.raises(IndexError):
        ops.check_embed_errors()


@pytest.mark.parametrize('act', ['none', 'relu'])
def test_linear_on_a_row_strided_view(ops, act):
    """The pooler's input hidden[:, 0] ([B, H] view of [B, L, H]: row stride L * H) goes to goat_gemm_bf16 as it is (lda = L * H) and is
    saved strided for the weight gradient (no op-level test covered forward / dW / dx on the strided operand)."""
    B, L, H, N = 48, 80, 768, 768
    g = torch.Generator().manual_seed(5)
    hid = torch.randn(B, L, H, generator=g).to(DEV, torch.bfloat16).requires_grad_(True)
    w = torch.nn.Parameter((torch.randn(N, H, generator=g) * 0.03).to(DEV))
    b = torch.nn.Parameter((torch.randn(N, generator=g) * 0.1).to(DEV))
    x = hid[:, 0]
    assert not x.is_contiguous() and x.stride(0) == L * H
    y = ops.linear(x, w, b, act=act) if act != 'none' else ops.linear(x, w, b)
    dy = torch.randn(B, N, generator=g).to(DEV, torch.bfloat16)
    y.backward(dy)
    hr = hid.detach().float().requires_grad_(True)
    wr, br = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = hr[:, 0] @ wr.to(torch.bfloat16).float().T + br
    if act == 'relu':
        yr = torch.relu(yr)
    yr.backward(dy.float())
    _close(y, yr, torch.bfloat16, 'strided linear y')
    _close(w.grad, wr.grad, torch.bfloat16, 'strided linear dW')
    _close(b.grad, br.grad, torch.bfloat16, 'strided linear db')
    _close(hid.grad[:, 0], hr.grad[:, 0], torch.bfloat16, 'strided linear dx')
    assert float(hid.grad[:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_linear_ffn_autograd(ops, dtype):
    M, H, F_ = 300, 768, 3072
    g = torch.Generator().manual_seed(21)
    x = torch.randn(M, H, generator=g).to(DEV, dtype).requires_grad_(True)
    w1 = torch.nn.Parameter((torch.randn(F_, H, generator=g) * 0.03).to(DEV))
    b1 = torch.nn.Parameter((torch.randn(F_, generator=g) * 0.1).to(DEV))
    w2 = torch.nn.Parameter((torch.randn(H, F_, generator=g) * 0.03).to(DEV))
    b2 = torch.nn.Parameter((torch.randn(H, generator=g) * 0.1).to(DEV))
    y = ops.ffn(x, w1, b1, w2, b2, 'gelu', 0.0)
    dy = torch.randn(M, H, generator=g).to(DEV, dtype)
    y.backward(dy)
    xr = x.detach().float().requires_grad_(True)
    ps = [p.detach().clone().requires_grad_(True) for p in (w1, b1, w2, b2)]
    if dtype == torch.bfloat16:     # the kernels see bf16-rounded weights
        wq = [ps[0].to(dtype).float(), ps[2].to(dtype).float()]
    else:
        wq = [ps[0], ps[2]]
    yr = torch.nn.functional.gelu(xr @ wq[0].T + ps[1]) @ wq[1].T + ps[3]
    yr.backward(dy.float())
    _close(y, yr, dtype, 'ffn y')
    _close(x.grad, xr.grad, dtype, 'ffn dx')
    for p, r, n in zip((w1, b1, w2, b2), ps, ('dw1', 'db1', 'dw2', 'db2')):
        _close(p.grad, r.grad, dtype, 'ffn ' + n)
    # small-K / N=1 linears go through the same GEMM
    xs = torch.randn(77, 7, generator=g).to(DEV, dtype).requires_grad_(True)
    ws = torch.nn.Parameter((torch.randn(768, 7, generator=g)).to(DEV))
    bs = torch.nn.Parameter(torch.randn(768, generator=g).to(DEV))
    ys = ops.linear(xs, ws, bs)
    ys.backward(torch.ones_like(ys))
    _close(ys, xs.float() @ (ws.to(dtype).float()).T + bs, dtype, 'K=7 linear')
    _close(ws.grad, torch.ones(77, 768, device=DEV).T @ xs.detach().float(), dtype, 'K=7 dW')
    _close(xs.grad, torch.ones(77, 768, device=DEV) @ ws.detach().to(dtype).float(), dtype, 'K=7 dx')


@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize('M,N,Kc', [(128, 128, 64), (3840, 768, 768), (1056, 768, 768), (200, 2304, 768), (37, 50, 128),
                                    (130, 1, 64), (768, 3072, 1003), (2304, 768, 1776), (64, 768, 48)])
def test_gemm_bf16_lds_dma_all_layouts(ops, ta, tb, M, N, Kc):
    """goat_gemm_bf16 (LDS-DMA pipeline, swizzled LDS images, tr-read transposed operands) vs torch fp32."""
    if not (ta and tb) and Kc % 64:
        pytest.skip('K-contiguous operands need Kc % 64 == 0 (routed to goat_gemm_nt by the dispatcher)')
    g = torch.Generator().manual_seed(M * 31 + N * 7 + Kc)
    A = torch.randn(M, Kc, generator=g)
    B = torch.randn(N, Kc, generator=g) * 0.1
    ld_pad = lambda n: (n + 7) // 8 * 8
    if ta:
        a = torch.zeros(Kc, ld_pad(M)); a[:, :M] = A.T
        a = a.to(DEV, torch.bfloat16)[:, :M]
    else:
        a = A.to(DEV, torch.bfloat16)
    if tb:
        b = torch.zeros(Kc, ld_pad(N)); b[:, :N] = B.T
        b = b.to(DEV, torch.bfloat16)[:, :N]
    else:
        b = B.to(DEV, torch.bfloat16)
    ref = (a.float().T if ta else a.float()) @ (b.float() if tb else b.float().T)
    bias = torch.randn(N, generator=g).to(DEV)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(a, b, out, ta=ta, tb=tb, bias=bias)
    _close(out, ref + bias, torch.bfloat16, 'gemm_bf16 bf16 out')
    out32 = torch.zeros(M, N, device=DEV, dtype=torch.float32)
    ops.gemm(a, b, out32, ta=ta, tb=tb)
    _close(out32, ref, torch.float32, 'gemm_bf16 f32 out') if False else _close(out32, ref, torch.bfloat16, 'f32 out')
    # split-K accumulates on top of the existing contents
    if (Kc + 63) // 64 >= 2:
        acc = torch.ones(M, N, device=DEV, dtype=torch.float32)
        ops.gemm(a, b, acc, ta=ta, tb=tb, split_k=3)
        _close(acc - 1.0, ref, torch.bfloat16, 'split-K')


def test_gemm_bf16_epilogues_match_v1(ops):
    from vln_goat_amd._lib import EPI_GELU, EPI_MUL_DGELU
    M, N, K = 500, 3072, 768
    g = torch.Generator().manual_seed(77)
    a = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(DEV, torch.bfloat16)
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    out, aux = torch.empty(M, N, device=DEV, dtype=torch.bfloat16), torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(a, w, out, bias=bias, epi=EPI_GELU, aux=aux)
    u = a.float() @ w.float().T + bias
    _close(aux, u, torch.bfloat16, 'aux')
    _close(out, torch.nn.functional.gelu(u), torch.bfloat16, 'gelu')
    dy = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w2 = (torch.randn(K, N, generator=g) * 0.05).to(DEV, torch.bfloat16)     # [Kc=K rows, N cols] -> tb
    du = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm(dy, w2, du, tb=True, epi=EPI_MUL_DGELU, aux=aux)
    uu = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(uu).sum().backward()
    _close(du, (dy.float() @ w2.float()) * uu.grad, torch.bfloat16, 'dgelu epilogue')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('M,N', [(37, 1000), (576, 50265)])
def test_decoder_cross_entropy_fused(ops, dtype, M, N):
    K = 768
    g = torch.Generator().manual_seed(N + M)
    h = torch.randn(M, K, generator=g).to(DEV, dtype).requires_grad_(True)
    w = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.05).to(DEV))
    b = torch.nn.Parameter((torch.randn(N, generator=g) * 0.1).to(DEV))
    tgt = torch.randint(0, N, (M,), generator=g).to(DEV)
    loss = ops.decoder_cross_entropy(h, w, b, tgt)
    gl = torch.rand(M, generator=g).to(DEV)
    loss.backward(gl)
    hr = h.detach().float().requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    wq = wr.to(dtype).float() if dtype == torch.bfloat16 else wr
    ref = torch.nn.functional.cross_entropy(hr @ wq.T + br, tgt, reduction='none')
    ref.backward(gl)
    _close(loss, ref, dtype, 'ce loss')
    _close(h.grad, hr.grad, dtype, 'ce dh')
    _close(w.grad, wr.grad, dtype, 'ce dW')
    _close(b.grad, br.grad, dtype, 'ce db')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('L', [22, 80, 200])
def test_attn_pool(ops, dtype, L):
    """tanh-attention pooling of the CFP heads (all slots, no mask) against the reference formula."""
    B, H = 5, 768
    g = torch.Generator().manual_seed(41)
    x = (torch.randn(B, L, H, generator=g) * 0.7).to(DEV, dtype).requires_grad_(True)
    w = torch.nn.Parameter((torch.rand(H, 1, generator=g) * 0.2 - 0.1).to(DEV))
    out = ops.attn_pool(x, w)
    dout = torch.randn(B, H, generator=g).to(DEV)
    out.backward(dout)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    a = torch.softmax(torch.matmul(torch.tanh(xr), wr), 1)
    ref = torch.tanh(torch.sum(xr * a, 1))
    ref.backward(dout)
    assert out.dtype == torch.float32
    _close(out, ref, dtype, 'attn_pool')
    _close(x.grad, xr.grad, dtype, 'attn_pool dx')
    _close(w.grad, wr.grad, dtype, 'attn_pool dw')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_door_gate(ops, dtype):
    rows, H = 333, 768
    g = torch.Generator().manual_seed(42)
    aug = torch.randn(3, 111, H, generator=g).to(DEV, dtype).requires_grad_(True)
    ori = torch.randn(3, 111, H, generator=g).to(DEV, dtype).requires_grad_(True)
    la, lo = torch.nn.Linear(H, 1).to(DEV), torch.nn.Linear(H, 1).to(DEV)
    out = ops.door_gate(la, lo, aug, ori)
    dout = torch.randn(3, 111, H, generator=g).to(DEV, dtype)
    out.backward(dout)
    got = [aug.grad, ori.grad, la.weight.grad.clone(), la.bias.grad.clone(), lo.weight.grad.clone(), lo.bias.grad.clone()]
    ar, orr = aug.detach().float().requires_grad_(True), ori.detach().float().requires_grad_(True)
    for m in (la, lo):
        m.zero_grad()
    s = torch.sigmoid(la(ar) + lo(orr))
    ref = s * ar + (1 - s) * orr
    ref.backward(dout.float())
    _close(out, ref, dtype, 'door')
    for a, b, n in zip(got, [ar.grad, orr.grad, la.weight.grad, la.bias.grad, lo.weight.grad, lo.bias.grad],
                       ['daug', 'dori', 'dwa', 'dba', 'dwo', 'dbo']):
        _close(a, b, dtype, 'door ' + n)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dict_weighted_sum(ops, dtype):
    B, K, H = 4, 39, 768
    g = torch.Generator().manual_seed(43)
    z = torch.rand(B, K, H, generator=g).to(DEV).requires_grad_(True)
    p = torch.rand(B, K, 1, generator=g).to(DEV).requires_grad_(True)
    out = ops.dict_weighted_sum(z, p, dtype)
    dout = torch.randn(B, 1, H, generator=g).to(DEV, dtype)
    out.backward(dout)
    zr, pr = z.detach().clone().requires_grad_(True), p.detach().clone().requires_grad_(True)
    ref = torch.sum(zr * pr, 1, keepdim=True)
    ref.backward(dout.float())
    assert out.shape == (B, 1, H) and out.dtype == dtype
    _close(out, ref, dtype, 'dict_wsum')
    _close(z.grad, zr.grad, dtype, 'dict_wsum dz')
    _close(p.grad, pr.grad, dtype, 'dict_wsum dp')


@pytest.mark.parametrize('ta,tb', [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize('bm,ns', [(64, 2), (64, 3), (64, 4), (128, 2), (128, 3), (128, 4), (256, 2), (256, 3),
                                   (128, 0x102), (128, 0x103), (128, 0x104),      # 0x100: the 128-row tile on eight waves
                                   (96, 2), (96, 3), (96, 4),                      # 96 x 128 (non-transposed A only)
                                   (128 | 256 << 16, 2), (128 | 256 << 16, 3), (192 | 256 << 16, 2), (256 | 192 << 16, 2),
                                   (256 | 256 << 16, 2), (192 | 192 << 16, 2), (192 | 192 << 16, 3),                          # rows | columns << 16: the 8-wave wide tiles
                                   (128, 0x202), (256, 0x202), (128 | 256 << 16, 0x202), (192 | 256 << 16, 0x202),
                                   (256 | 256 << 16, 0x202)])                      # 0x200: the ping-pong main loop (csrc/gemm5_tile.hpp)
def test_gemm_bf16_every_tile_configuration(ops, ta, tb, bm, ns):
    """Every (tile, ring depth) the autotuner may pick, on a ragged shape, incl. GELU / x GELU' / C += A·B / split-K epilogues."""
    from vln_goat_amd._lib import EPI_ACCUM, EPI_GELU, EPI_MUL_DGELU, EPI_NONE
    rows, cols = bm & 0xFFFF, (bm >> 16) or 128
    if (ta and rows & (rows - 1)) or (tb and cols & (cols - 1)):
        pytest.skip('a transposed operand needs a power-of-two tile width on its side')
    M, N, Kc = 1000, 392, 320 if not (ta and tb) else 328
    g = torch.Generator().manual_seed(bm * 7 + ns)
    A = torch.randn(M, Kc, generator=g)
    B = torch.randn(N, Kc, generator=g) * 0.1
    a = (A.T.contiguous() if ta else A).to(DEV, torch.bfloat16)
    b = (B.T.contiguous() if tb else B).to(DEV, torch.bfloat16)
    ref = (a.float().T if ta else a.float()) @ (b.float() if tb else b.float().T)
    bias = torch.randn(N, generator=g).to(DEV)

    def run(out, epi=EPI_NONE, aux=None, split=1, bias_=None):
        ops._launch_gemm_bf16(a, b, out, ta, tb, M, N, Kc, bias_, epi, aux, split, bm, ns, None)
        return out
    out = run(torch.empty(M, N, device=DEV, dtype=torch.bfloat16), bias_=bias)
    _close(out, ref + bias, torch.bfloat16, 'plain')
    if not (bm == 128 and (ns & 0xFF) == 4) and not ta:      # (activation epilogues: forward / dgrad layouts; the 128-row tile has them for 2-3 ring slots)
        aux = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        out = run(torch.empty(M, N, device=DEV, dtype=torch.bfloat16), EPI_GELU, aux, bias_=bias)
        _close(aux, ref + bias, torch.bfloat16, 'gelu pre-activation')
        _close(out, torch.nn.functional.gelu(ref + bias), torch.bfloat16, 'gelu')
        uu = aux.float().requires_grad_(True)
        torch.nn.functional.gelu(uu).sum().backward()
        out = run(torch.empty(M, N, device=DEV, dtype=torch.bfloat16), EPI_MUL_DGELU, aux)
        _close(out, ref * uu.grad, torch.bfloat16, "x gelu'")
    out32 = run(torch.empty(M, N, device=DEV), bias_=bias)
    _close(out32, ref + bias, torch.bfloat16, 'f32 result with bias')
    acc = run(torch.full((M, N), 2.0, device=DEV), EPI_ACCUM)
    _close(acc - 2.0, ref, torch.bfloat16, 'accumulate')
    acc = run(torch.full((M, N), -1.0, device=DEV), split=3)
    _close(acc + 1.0, ref, torch.bfloat16, 'split-K')


@pytest.mark.parametrize('tb', [False, True])
@pytest.mark.parametrize('bm', [128, 256, 128 | 256 << 16, 192 | 256 << 16, 256 | 256 << 16])
def test_gemm_bf16_persistent_matches_one_workgroup_per_tile(ops, bm, tb):
    """GOAT_GEMM_PERSIST (0x400, with the ping-pong loop): one workgroup per CU walks several tiles and requests the next tile's first
    K-tile before the current epilogue.  Results must be BIT-identical to the one-workgroup-per-tile launch, with more tiles than CUs
    (ragged edges included) and with fewer; bias, GELU (+ saved pre-activation) and x GELU' epilogues."""
    from vln_goat_amd._lib import EPI_GELU, EPI_MUL_DGELU, EPI_NONE
    rows, cols = bm & 0xFFFF, (bm >> 16) or 128
    if tb and cols & (cols - 1):
        pytest.skip('a transposed operand needs a power-of-two tile width on its side')
    for M, N, Kc in ((rows * 37 + 5, cols * 9 - 24, 320), (rows * 3, cols * 2, 768), (rows * 23, cols * 12, 64)):
        g = torch.Generator().manual_seed(bm + M)
        a = torch.randn(M, Kc, generator=g).to(DEV, torch.bfloat16)
        B = torch.randn(N, Kc, generator=g) * 0.1
        b = (B.T.contiguous() if tb else B).to(DEV, torch.bfloat16)
        bias = torch.randn(N, generator=g).to(DEV)

        def run(ns, epi=EPI_NONE, aux=None, bias_=None):
            out = torch.full((M, N), 3.0, device=DEV, dtype=torch.bfloat16)
            ops._launch_gemm_bf16(a, b, out, False, tb, M, N, Kc, bias_, epi, aux, 1, bm, ns, None)
            return out
        ref = run(0x202, bias_=bias)
        got = run(0x602, bias_=bias)
        assert torch.equal(ref, got), ('plain', M, N, Kc, float((ref.float() - got.float()).abs().max()))
        _close(got, a.float() @ (b.float() if tb else b.float().T) + bias, torch.bfloat16, 'persistent vs torch')
        aux0, aux1 = (torch.zeros(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(2))
        assert torch.equal(run(0x202, EPI_GELU, aux0, bias), run(0x602, EPI_GELU, aux1, bias)) and torch.equal(aux0, aux1)
        assert torch.equal(run(0x202, EPI_MUL_DGELU, aux0), run(0x602, EPI_MUL_DGELU, aux0))


def test_wgrad_grouped(ops):
    """goat_wgrad_grouped: several dW_i = dY_i^T · X_i problems of different shapes (ragged rows, row-strided dY) in one
    launch, with bias column sums; both tile heights."""
    import ctypes
    from vln_goat_amd import _lib
    g = torch.Generator().manual_seed(77)
    shapes = [(3840, 768, 768), (1000, 2304, 768), (333, 768, 3072), (3840, 3072, 768), (37, 8, 768), (576, 1001, 768)]
    for bm, ns in ((64, 3), (128, 2), (128, 0x102), (128, 0x103), (256, 2), (128 | 256 << 16, 3), (256 | 256 << 16, 2),
                   (256, 0x202), (128 | 256 << 16, 0x202), (256 | 256 << 16, 0x202)):
        # 0x100: eight waves on the 128-row tile; rows | columns << 16; 0x200: ping-pong main loop
        arr = (_lib.WgradProblem * len(shapes))()
        keep, refs = [], []
        for i, (rows, n_out, n_in) in enumerate(shapes):
            ld = (n_out + 7) // 8 * 8 + 8                                   # row-strided dY (a column slice of a wider buffer)
            dyb = (torch.randn(rows, ld, generator=g) * 0.5).to(DEV, torch.bfloat16)
            dy = dyb[:, :n_out]
            x = torch.randn(rows, n_in, generator=g).to(DEV, torch.bfloat16)
            dw = torch.full((n_out, n_in), 7.0, device=DEV)                  # stale contents: the launch overwrites
            db = torch.zeros(n_out, device=DEV)
            q = arr[i]
            q.dy, q.ld_dy, q.x, q.ld_x, q.dw, q.ld_dw, q.dbias = dy.data_ptr(), ld, x.data_ptr(), n_in, dw.data_ptr(), n_in, db.data_ptr()
            q.rows, q.n_out, q.n_in, q.accumulate = rows, n_out, n_in, 0
            keep.append((dyb, x, dw, db))
            refs.append((dy.float().T @ x.float(), dy.float().sum(0)))
        st = _lib.lib().goat_wgrad_grouped(torch.cuda.current_stream().cuda_stream, ctypes.addressof(arr), len(shapes), bm, ns)
        assert st == 0
        for (_, _, dw, db), (rw, rb) in zip(keep, refs):
            _close(dw, rw, torch.bfloat16, 'grouped dW')
            _close(db, rb, torch.bfloat16, 'grouped dbias')
        # accumulate flag: second launch adds on top
        for i in range(len(shapes)):
            arr[i].accumulate = 1
            arr[i].dbias = None
        st = _lib.lib().goat_wgrad_grouped(torch.cuda.current_stream().cuda_stream, ctypes.addressof(arr), len(shapes), bm, ns)
        assert st == 0
        for (_, _, dw, _), (rw, _) in zip(keep, refs):
            _close(dw, 2 * rw, torch.bfloat16, 'grouped dW accumulate')


def test_wgrad_grouped_forty_eight_problems(ops):
    """The argument block of a grouped launch holds 48 problems (64-byte records expanded on the device): 48 problems of mixed shapes in
    one launch, per-tile and balanced forms; a 49th is refused."""
    import ctypes
    from vln_goat_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(79)
    shapes = [((200 + 37 * i) % 900 + 64, (96 * (i % 5) + 40), (128 * (i % 4) + 72)) for i in range(48)]
    arr = (_lib.WgradProblem * 49)()
    keep, refs = [], []
    for i, (rows, n_out, n_in) in enumerate(shapes + [shapes[0]]):
        ld = (n_out + 7) // 8 * 8
        dyb = (torch.randn(rows, ld, generator=g) * 0.5).to(DEV, torch.bfloat16)
        dy = dyb[:, :n_out]
        ldx = (n_in + 7) // 8 * 8
        xb = torch.randn(rows, ldx, generator=g).to(DEV, torch.bfloat16)
        x = xb[:, :n_in]
        dw = torch.full((n_out, n_in), 7.0, device=DEV)
        db = torch.zeros(n_out, device=DEV)
        q = arr[i]
        q.dy, q.ld_dy, q.x, q.ld_x, q.dw, q.ld_dw, q.dbias = dy.data_ptr(), ld, x.data_ptr(), ldx, dw.data_ptr(), n_in, db.data_ptr()
        q.rows, q.n_out, q.n_in, q.accumulate = rows, n_out, n_in, 0
        keep.append((dyb, xb, dw, db))
        refs.append((dy.float().T @ x.float(), dy.float().sum(0)))
    st = torch.cuda.current_stream().cuda_stream
    for bm, ns in ((256 | 256 << 16, 0x202), (256, 3), (128, 0x102)):
        for (_, _, dw, db) in keep:
            dw.fill_(7.0)
            db.zero_()
        assert L.goat_wgrad_grouped(st, ctypes.addressof(arr), 48, bm, ns) == 0
        for (_, _, dw, db), (rw, rb) in list(zip(keep, refs))[:48]:
            _close(dw, rw, torch.bfloat16, '48 problems dW')
            _close(db, rb, torch.bfloat16, '48 problems dbias')
    bm = 256 | 256 << 16
    nb = L.goat_wgrad_balanced_ws_bytes(bm)
    ws = torch.zeros(nb, dtype=torch.uint8, device=DEV)
    for (_, _, dw, db) in keep:
        dw.fill_(7.0)
        db.zero_()
    assert L.goat_wgrad_grouped_balanced(st, ctypes.addressof(arr), 48, bm, ws.data_ptr(), nb) == 0
    for (_, _, dw, db), (rw, rb) in list(zip(keep, refs))[:48]:
        _close(dw, rw, torch.bfloat16, '48 problems dW (balanced)')
        _close(db, rb, torch.bfloat16, '48 problems dbias (balanced)')
    assert L.goat_wgrad_grouped(st, ctypes.addressof(arr), 49, bm, 0x202) != 0


def test_wgrad_grouped_balanced(ops):
    """goat_wgrad_grouped_balanced (one workgroup per CU, equal shares of the group's K-tile iterations, cut tiles summed through the
    workspace): mixed contraction lengths so that shares hold tails, whole tiles and heads, and tiles whose contraction spans three
    workgroups (rows 40000 on few tiles); ragged edges, row-strided dY, bias sums.  Against float32 torch, against the one-workgroup-
    per-tile launch, bit-identical from launch to launch (flags return to zero), accumulate on top of stale contents."""
    import ctypes
    from vln_goat_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(78)
    cases = [[(3840, 768, 768), (1000, 2304, 768), (8640, 768, 3072), (3840, 3072, 768), (37, 8, 768), (576, 1001, 768), (8640, 2304, 768)],
             [(40000, 256, 512), (40000, 300, 256), (640, 768, 768)],
             [(1776, o, i) for (o, i) in ((2304, 768), (768, 768), (3072, 768), (768, 3072))] + [(1056, 768, 768), (1056, 3072, 768)]]
    st = torch.cuda.current_stream().cuda_stream
    for shapes in cases:
        for bm in (256 | 256 << 16, 128 | 256 << 16, 256):
            nb = L.goat_wgrad_balanced_ws_bytes(bm)
            assert nb > 0
            ws = torch.zeros(nb, dtype=torch.uint8, device=DEV)
            arr = (_lib.WgradProblem * len(shapes))()
            keep, refs = [], []
            for i, (rows, n_out, n_in) in enumerate(shapes):
                ld = (n_out + 7) // 8 * 8 + 8
                dyb = (torch.randn(rows, ld, generator=g) * 0.5).to(DEV, torch.bfloat16)
                dy = dyb[:, :n_out]
                x = torch.randn(rows, n_in, generator=g).to(DEV, torch.bfloat16)
                dw = torch.full((n_out, n_in), 7.0, device=DEV)
                db = torch.zeros(n_out, device=DEV)
                q = arr[i]
                q.dy, q.ld_dy, q.x, q.ld_x, q.dw, q.ld_dw, q.dbias = dy.data_ptr(), ld, x.data_ptr(), n_in, dw.data_ptr(), n_in, db.data_ptr()
                q.rows, q.n_out, q.n_in, q.accumulate = rows, n_out, n_in, 0
                keep.append((dyb, x, dw, db))
                refs.append((dy.float().T @ x.float(), dy.float().sum(0)))
            assert L.goat_wgrad_grouped_balanced(st, ctypes.addressof(arr), len(shapes), bm, ws.data_ptr(), nb) == 0
            first = [(dw.clone(), db.clone()) for (_, _, dw, db) in keep]
            for (dw, db), (rw, rb) in zip(first, refs):
                _close(dw, rw, torch.bfloat16, 'balanced dW')
                _close(db, rb, torch.bfloat16, 'balanced dbias')
            assert int(ws[-256 * 32:].view(torch.int32).abs().sum()) == 0, 'flags must return to zero'
            # the per-tile launch on the same operands: same numbers up to float32 summation order
            for (_, _, dw, db) in keep:
                dw.fill_(7.0)
                db.zero_()
            assert L.goat_wgrad_grouped(st, ctypes.addressof(arr), len(shapes), bm, 0x202) == 0
            for (dw0, db0), (_, _, dw, db) in zip(first, keep):
                assert float((dw0 - dw).abs().max()) <= 1e-4 * max(1.0, float(dw.abs().max())), 'balanced vs per-tile'
                assert float((db0 - db).abs().max()) <= 1e-4 * max(1.0, float(db.abs().max()))
            # deterministic: five more launches reproduce the first bit for bit
            for _ in range(5):
                for (_, _, dw, db) in keep:
                    dw.fill_(-3.0)
                    db.zero_()
                assert L.goat_wgrad_grouped_balanced(st, ctypes.addressof(arr), len(shapes), bm, ws.data_ptr(), nb) == 0
                for (dw0, db0), (_, _, dw, db) in zip(first, keep):
                    assert torch.equal(dw0, dw) and torch.equal(db0, db)
            for i in range(len(shapes)):
                arr[i].accumulate = 1
                arr[i].dbias = None
            assert L.goat_wgrad_grouped_balanced(st, ctypes.addressof(arr), len(shapes), bm, ws.data_ptr(), nb) == 0
            for (_, _, dw, _), (rw, _) in zip(keep, refs):
                _close(dw, 2 * rw, torch.bfloat16, 'balanced dW accumulate')
    # fewer iterations than CUs: refused, the caller uses the per-tile launch
    arr = (_lib.WgradProblem * 1)()
    dy, x, dw = torch.zeros(64, 256, device=DEV, dtype=torch.bfloat16), torch.zeros(64, 256, device=DEV, dtype=torch.bfloat16), torch.zeros(256, 256, device=DEV)
    q = arr[0]
    q.dy, q.ld_dy, q.x, q.ld_x, q.dw, q.ld_dw, q.dbias = dy.data_ptr(), 256, x.data_ptr(), 256, dw.data_ptr(), 256, None
    q.rows, q.n_out, q.n_in, q.accumulate = 64, 256, 256, 0
    bm = 256 | 256 << 16
    nb = L.goat_wgrad_balanced_ws_bytes(bm)
    ws = torch.zeros(nb, dtype=torch.uint8, device=DEV)
    assert L.goat_wgrad_grouped_balanced(st, ctypes.addressof(arr), 1, bm, ws.data_ptr(), nb) == -2
    assert L.goat_wgrad_grouped_balanced(st, ctypes.addressof(arr), 1, bm, ws.data_ptr(), nb - 1) != 0
    assert L.goat_wgrad_balanced_ws_bytes(64) < 0


def test_gemm_bf16_full_size_every_tile_no_corruption(ops):
    """Full-size FFN-up shape (3840 x 3072 x 768, GELU + saved pre-activation) on every tile the autotuner may pick, launched
    several times: no NaN / wrong element anywhere.  (Round 2 found isolated wrong elements on the 4-wave 128x128 tile only at
    this size: the inline-asm non-temporal store lacked the `s_nop` that covers the >64-bit store-data hazard.)"""
    from vln_goat_amd._lib import EPI_GELU
    M, N, K = 3840, 3072, 768
    g = torch.Generator().manual_seed(3)
    a = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.05).to(DEV, torch.bfloat16)
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    u = a.float() @ b.float().T + bias
    want = torch.nn.functional.gelu(u)
    scale = float(want.abs().max())
    for bm, ns in ops._tile_candidates(False, False, M, N):
        out = torch.full((M, N), float('nan'), device=DEV, dtype=torch.bfloat16)
        aux = torch.full((M, N), float('nan'), device=DEV, dtype=torch.bfloat16)
        try:
            for _ in range(3):
                ops._launch_gemm_bf16(a, b, out, False, False, M, N, K, bias, EPI_GELU, aux, 1, bm, ns, None)
        except RuntimeError:
            continue                                   # (a configuration this build does not instantiate)
        torch.cuda.synchronize()
        assert not bool(torch.isnan(out.float()).any()) and not bool(torch.isnan(aux.float()).any()), (bm, ns)
        assert float((out.float() - want).abs().max()) < 2e-2 * scale, (bm, ns)
        assert float((aux.float() - u).abs().max()) < 2e-2 * float(u.abs().max()), (bm, ns)


@pytest.mark.parametrize('Bl,Ba,t0', [(48, 48, 0), (5, 5, 0), (6, 18, 6)])
def test_infonce_fused_matches_torch(ops, Bl, Ba, t0):
    """goat_infonce_fwd / _bwd against the torch formulation of P/model/pretrain_goat.py:519-534 — one rank (loc == all: the same
    tensors, gradients of both roles land in one buffer) and the data-parallel layout (all = gathered rows, target offset)."""
    H, tau = 768, 0.7
    g = torch.Generator().manual_seed(Bl * 31 + Ba)
    full = [torch.tanh(torch.randn(Ba, H, generator=g)).to(DEV) for _ in range(4)]
    if Ba == Bl:
        loc = [t.clone().requires_grad_(True) for t in full]
        alls = loc
    else:
        loc = [t[t0:t0 + Bl].clone().requires_grad_(True) for t in full]
        alls = [t.clone().requires_grad_(True) for t in full]
    loss = ops.infonce(loc[0], loc[1], loc[2], loc[3], alls[0], alls[1], alls[2], alls[3], t0, tau)
    w = torch.rand(Bl, generator=g).to(DEV)
    (loss * w).sum().backward()
    rl = [t.detach().clone().requires_grad_(True) for t in loc]
    ra = rl if Ba == Bl else [t.detach().clone().requires_grad_(True) for t in alls]
    tgt = torch.arange(Bl, device=DEV) + t0
    ref = 0
    for k in range(3):
        row = torch.nn.functional.cross_entropy((rl[k] @ ra[3].T) / tau, tgt, reduction='none')
        col = torch.nn.functional.cross_entropy((rl[3] @ ra[k].T) / tau, tgt, reduction='none')
        ref = ref + (row + col) / 2.0
    (ref * w).sum().backward()
    _close(loss, ref, torch.float32, 'infonce loss')
    for k in range(4):
        _close(loc[k].grad, rl[k].grad, torch.float32, 'infonce dloc %d' % k)
        if Ba != Bl:
            _close(alls[k].grad, ra[k].grad, torch.float32, 'infonce dall %d' % k)


def test_grouped_wgrad_tail_split_plan(ops):
    """WgradQueue plans: a group whose 256x128 tiles leave a mostly empty last round is run as two launches (the tail problems on
    128x128 tiles); results and accumulate flags are those of the single launch."""
    W = ops.WgradQueue
    g = torch.Generator().manual_seed(12)
    layer = [(2304, 768), (768, 768), (3072, 768), (768, 3072)]
    rows = 640
    q, refs = [], []
    for rep in range(2):
        for n_out, n_in in layer:
            dy = (torch.randn(rows, n_out, generator=g) * 0.3).to(DEV, torch.bfloat16)
            x = torch.randn(rows, n_in, generator=g).to(DEV, torch.bfloat16)
            acc = rep                                        # second layer's problems accumulate onto existing contents
            w = torch.full((n_out, n_in), 2.0 if acc else 9.0, device=DEV)
            b = torch.zeros(n_out, device=DEV)
            q.append((dy, x, w, b, acc))
            refs.append((dy.float().T @ x.float() + (2.0 if acc else 0.0), dy.float().sum(0)))
    tail = W._tail_split(q)
    assert tail is not None and 0 < len(tail) < len(q)
    key = tuple((t[0].shape[0], t[0].shape[1], t[1].shape[1]) for t in q)
    head = tuple(i for i in range(len(q)) if i not in tail)
    keep_env = os.environ.pop('GOAT_WGRAD_GROUP_CFG', None)
    try:
        W.tuned[key] = [(head, (256, 3)), (tail, (128, ops.EIGHT_WAVES | 2))]
        W._launch(q)
        torch.cuda.synchronize()
    finally:
        W.tuned.pop(key, None)
        if keep_env is not None:
            os.environ['GOAT_WGRAD_GROUP_CFG'] = keep_env
    for (dy, x, w, b, acc), (rw, rb) in zip(q, refs):
        _close(w, rw, torch.bfloat16, 'tail-split dW')
        _close(b, rb, torch.bfloat16, 'tail-split dbias')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('form,B,G,W', [('pretrain', 6, 23, 38), ('nav', 6, 23, 38), ('nav', 6, 150, 71), ('pretrain', 6, 65, 1)])
def test_sap_fuse_matches_the_reference_chain(ops, dtype, form, B, G, W):
    """hipops.sap_fuse (one launch per direction) against the reference's spelling of the SAP head tail with torch ops: scaled by the
    fusion weight, masked_fill x4, bmm with the logit-fusion matrix, (fine-tuning form: local stop logit added to the stop column),
    three cross-entropies — P/model/pretrain_goat.py:375-413, M/models/vilmodel_GOAT.py:803-839.  Logits, loss and the gradients of
    both score tensors and of the fusion logit; also with upstream gradients on the logits (no labels) and without a fusion Linear."""
    g = torch.Generator().manual_seed(11)
    gs0 = torch.randn(B, G, generator=g).to(DEV, dtype)
    ls0 = torch.randn(B, W, generator=g).to(DEV, dtype)
    fwl0 = torch.randn(B, 1, generator=g).to(DEV, dtype)
    glens = torch.tensor([G, 10, 17, 5, G, 12], device=DEV).clamp(max=G)      # (maps longer than a wave: the kernel loops)
    valid = torch.arange(G, device=DEV)[None, :] < glens[:, None]
    vis = (torch.rand(B, G, generator=g) < 0.3).to(DEV) & valid
    vis[:, 0] = False
    lvalid = (torch.rand(B, W, generator=g) < 0.6).to(DEV)
    lvalid[:, 0] = True
    M = (torch.rand(B, G, W, generator=g) < 0.05).float().to(DEV)
    ga = torch.tensor([0, 3, 5, 1, 7, 2], device=DEV)
    for b in range(B):                                            # labels on unmasked slots
        ok = (~vis[b] & valid[b]).nonzero().flatten()
        ga[b] = ok[int(ga[b]) % len(ok)]
    la = torch.stack([lvalid[b].nonzero().flatten()[b % int(lvalid[b].sum())] for b in range(B)])
    nav = form == 'nav'

    def reference(gs, ls, fwl):
        fw = 0.5 if fwl is None else torch.sigmoid(fwl.float())
        gl = gs.float() * fw
        ll = ls.float() * (1 - fw)
        gl = gl.masked_fill(vis, -float('inf')).masked_fill(valid.logical_not(), -float('inf'))
        navm = lvalid.logical_not()
        ll = ll.masked_fill(navm, -float('inf'))
        fused = gl.clone()
        if nav:
            add0 = torch.zeros_like(fused)
            add0[:, 0] = ll[:, 0]
            fused = fused + add0
        fused = fused + torch.bmm(M, ll.masked_fill(navm, 0.0).unsqueeze(2)).squeeze(2)
        return gl, ll, fused

    F = torch.nn.functional
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    for with_fw in (True, False):
        for labels in (True, False):
            leaves = [gs0.clone().requires_grad_(True), ls0.clone().requires_grad_(True), fwl0.clone().requires_grad_(True) if with_fw else None]
            rl, rll, rf = reference(*leaves)
            wg, wl, wf = (torch.randn(B, G, generator=g).to(DEV), torch.randn(B, W, generator=g).to(DEV), torch.randn(B, G, generator=g).to(DEV))
            fin = lambda t, w: (torch.where(torch.isfinite(t), t, torch.zeros_like(t)) * w).sum()
            if labels:
                rloss = F.cross_entropy(rl, ga, reduction='none') + F.cross_entropy(rll, la, reduction='none') + F.cross_entropy(rf, ga, reduction='none')
                (rloss * torch.arange(1, B + 1, device=DEV)).sum().backward()
            else:
                (fin(rl, wg) + fin(rll, wl) + fin(rf, wf)).backward()
            rgrads = [t.grad.float() if t is not None else None for t in leaves]
            mine = [gs0.clone().requires_grad_(True), ls0.clone().requires_grad_(True), fwl0.clone().requires_grad_(True) if with_fw else None]
            kw = dict(gvis=vis, lmask=lvalid, lmask_is_valid=True, M=M, add_stop=nav)
            kw.update(dict(gvalid=valid) if nav else dict(glens=glens))
            gl, ll, fu, loss = ops.sap_fuse(mine[0], mine[1], mine[2], labels=(ga, la) if labels else None, **kw)
            for a, r in ((gl, rl), (ll, rll), (fu, rf)):
                assert torch.equal(torch.isfinite(a), torch.isfinite(r))
                m = torch.isfinite(r)
                assert float((a[m] - r[m]).abs().max()) <= tol * max(1.0, float(r[m].abs().max()))
            if labels:
                assert float((loss - rloss).abs().max()) <= tol * max(1.0, float(rloss.abs().max()))
                (loss * torch.arange(1, B + 1, device=DEV)).sum().backward()
            else:
                assert loss is None
                (fin(gl, wg) + fin(ll, wl) + fin(fu, wf)).backward()
            torch.cuda.synchronize()
            for t, r, what in zip(mine, rgrads, ('d scores global', 'd scores local', 'd fusion logit')):
                if r is None:
                    continue
                assert float((t.grad.float() - r).abs().max()) <= 2 * tol * max(1.0, float(r.abs().max())), (what, with_fw, labels)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('p', [0.0, 0.3])
def test_layer_norm_z_out_fuses_the_residual_junction_in_front(ops, dtype, p):
    """hipops.layer_norm(x, residual, p, z_out=True) -> (LayerNorm(z), z), z = residual + dropout_p(x): the junction of a pre-LN block
    (P/model/transformer.py:172-176) and the LayerNorm behind it in one launch per direction.  z is used twice downstream (skip
    connection + through the norm): its second gradient joins inside the backward kernel (GOAT_LN_ADD_BEFORE).  Checked against
    torch with the dropout mask recovered from the zeros of dx."""
    g = torch.Generator().manual_seed(9)
    M, H = 300, 768
    x0 = torch.randn(2, M // 2, H, generator=g).to(DEV, dtype)
    r0 = torch.randn(2, M // 2, H, generator=g).to(DEV, dtype)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.1 * torch.randn(H, generator=g)).to(DEV).requires_grad_(True)
    wy = torch.randn(2, M // 2, H, generator=g).to(DEV)
    wz = torch.randn(2, M // 2, H, generator=g).to(DEV)
    x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
    ops.manual_seed(77)
    y, z = ops.layer_norm(x, gamma, beta, 1e-5, r, p, z_out=True)
    ((y.float() * wy).sum() + (z.float() * wz).sum()).backward()
    got = [t.grad.float().clone() for t in (x, r, gamma, beta)]
    # reference with the same mask
    # (a dropped element receives exactly zero gradient; a kept one a non-zero one up to measure-zero coincidences)
    keep = torch.ones(x0.shape, dtype=torch.bool, device=DEV) if p == 0 else (got[0] != 0)
    scale = 1.0 / (1.0 - p)
    if p > 0:
        frac = float(keep.float().mean())
        assert abs(frac - (1 - p)) < 0.02
    xr, rr = x0.float().clone().requires_grad_(True), r0.float().clone().requires_grad_(True)
    g2, b2 = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
    zr = rr + torch.where(keep, xr * scale, torch.zeros_like(xr))
    yr = torch.nn.functional.layer_norm(zr, (H,), g2, b2, 1e-5)
    ((yr * wy).sum() + (zr * wz).sum()).backward()
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert float((z.float() - zr).abs().max()) <= tol * max(1.0, float(zr.abs().max()))
    assert float((y.float() - yr).abs().max()) <= tol * max(1.0, float(yr.abs().max()))
    for a, b, what in zip(got, (xr.grad, rr.grad, g2.grad, b2.grad), ('dx', 'dresidual', 'dgamma', 'dbeta')):
        assert float((a - b).abs().max()) <= 2 * tol * max(1.0, float(b.abs().max())), what


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_layer_norm_output_dropout_in_the_same_launch(ops, dtype):
    """hipops.layer_norm(..., p_out): y = dropout(LayerNorm(x)) (the embedding blocks, P/model/Bert_backbone.py:108-110) in one launch per
    direction; mask recovered from the zeros of y, forward and all gradients against torch with that mask; keep rate checked."""
    g = torch.Generator().manual_seed(4)
    M, H, p = 260, 768, 0.25
    x0 = torch.randn(M, H, generator=g).to(DEV, dtype)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV).requires_grad_(True)
    beta = (0.5 + 0.1 * torch.randn(H, generator=g)).to(DEV).requires_grad_(True)
    w = torch.randn(M, H, generator=g).to(DEV)
    for with_post in (False, True):
        # post_add: dropout(LayerNorm(x) + post) — the image embedding block's sum of two normalised tensors (P/model/vilmodel_goat.py:340-344)
        x = x0.clone().requires_grad_(True)
        post = (2 + torch.randn(M, H, generator=g)).to(DEV, dtype).requires_grad_(True) if with_post else None
        gamma.grad = beta.grad = None
        ops.manual_seed(5)
        y = ops.layer_norm(x, gamma, beta, 1e-12, p_out=p, post_add=post)
        (y.float() * w).sum().backward()
        keep = y.detach() != 0
        assert abs(float(keep.float().mean()) - (1 - p)) < 0.02
        xr = x0.float().clone().requires_grad_(True)
        pr = post.detach().float().clone().requires_grad_(True) if with_post else None
        g2, b2 = gamma.detach().clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
        pre = torch.nn.functional.layer_norm(xr, (H,), g2, b2, 1e-12)
        pre = pre + pr if with_post else pre
        yr = torch.where(keep, pre / (1 - p), torch.zeros_like(xr))
        (yr * w).sum().backward()
        tol = 1e-4 if dtype == torch.float32 else 2e-2
        assert float((y.float() - yr).abs().max()) <= tol * max(1.0, float(yr.abs().max()))
        pairs = [(x.grad.float(), xr.grad, 'dx'), (gamma.grad, g2.grad, 'dgamma'), (beta.grad, b2.grad, 'dbeta')]
        if with_post:
            pairs.append((post.grad.float(), pr.grad, 'dpost'))
        for a, b, what in pairs:
            assert float((a - b).abs().max()) <= 2 * tol * max(1.0, float(b.abs().max())), (what, with_post)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n,numel', [(2, 4096), (6, 48 * 80 * 768), (8, 1003), (3, 7)])
def test_add_n_and_fanout(ops, dtype, n, numel):
    """goat_add_n: the gradient fan-in of a tensor with n consumers in one launch (float32 accumulation) == the sum of the parts;
    hipops.fanout: n handles on one buffer whose gradients meet in that launch == plain autograd with n uses of the tensor."""
    torch.manual_seed(n * 1000 + numel % 97)
    parts = [torch.randn(numel, device=DEV).to(dtype) for _ in range(n)]
    ref = torch.stack([p.float() for p in parts]).sum(0)
    got = ops.add_n(parts)
    assert got.dtype == dtype
    _close(got, ref, dtype, 'add_n')
    if dtype == torch.float32:
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # in place on the first summand
    first = parts[0].clone()
    ops.add_n([first] + parts[1:], out=first)
    assert torch.equal(first, got)
    # fanout vs. n plain uses
    x = torch.randn(5, 13, device=DEV).to(dtype).requires_grad_(True)
    ws = [torch.randn(5, 13, device=DEV).to(dtype) for _ in range(n)]
    hs = ops.fanout(x, n)
    assert len(hs) == n and all(h.data_ptr() == x.data_ptr() for h in hs)
    sum((h * w).sum() for h, w in zip(hs[:-1], ws[:-1])).backward()          # the last handle stays unused: its gradient is None
    g_fan = x.grad.clone()
    x.grad = None
    sum((x * w).sum() for w in ws[:-1]).backward()
    _close(g_fan, x.grad, dtype, 'fanout gradient')


def test_zero_ranges_clears_exactly_the_ranges(ops):
    """goat_zero_ranges: the arena's per-step fills as one launch; neighbours of the ranges keep their contents."""
    buf = torch.full((1 << 16,), 3.0, device=DEV)
    views = [buf[64:192], buf[1024:1024 + 4096], buf[40000:40004], buf[50000:50000 + 12 * 1024]]
    ops.zero_ranges(views)
    ref = torch.full((1 << 16,), 3.0)
    for a, b in ((64, 192), (1024, 1024 + 4096), (40000, 40004), (50000, 50000 + 12 * 1024)):
        ref[a:b] = 0
    assert torch.equal(buf.cpu(), ref)
    many = [buf[i * 256:i * 256 + 64] for i in range(40)]                    # more than 16 ranges: several launches
    buf.fill_(5.0)
    ops.zero_ranges(many)
    ref = torch.full((1 << 16,), 5.0)
    for i in range(40):
        ref[i * 256:i * 256 + 64] = 0
    assert torch.equal(buf.cpu(), ref)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_linear_accepts_inputs_already_padded_along_k(ops, dtype):
    """train_step.prepare_position_features hands the 7- / 14-wide features over cast and zero-padded to the GEMM's K chunk:
    same outputs and parameter gradients as the unpadded input."""
    torch.manual_seed(3)
    for K in (7, 14):
        w = (torch.randn(768, K, device=DEV) * 0.1).requires_grad_(True)
        b = torch.randn(768, device=DEV).requires_grad_(True)
        x = torch.randn(50, 9, K, device=DEV)
        e = 8 if dtype == torch.bfloat16 else 4
        xp = torch.nn.functional.pad(x.to(dtype), (0, (-K) % e))
        y0 = ops.linear(x.to(dtype), w, b)
        g = torch.randn_like(y0)
        y0.backward(g)
        gw0, gb0 = w.grad.clone(), b.grad.clone()
        w.grad = b.grad = None
        y1 = ops.linear(xp, w, b)
        y1.backward(g)
        assert torch.equal(y0, y1)
        _close(w.grad, gw0, dtype, 'dW')
        _close(b.grad, gb0, dtype, 'db')
    with pytest.raises(ValueError):
        ops.linear(torch.randn(4, 11, device=DEV).to(dtype), torch.randn(8, 7, device=DEV))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_cfp_mix_and_tail_match_the_torch_chain(ops, dtype):
    """goat_cfp_mix_fwd/_bwd (fo = go w + vo (1 - w), w = sigmoid(fusion logit)) and hipops.cfp_tail (mix + the three symmetric InfoNCE losses
    as one autograd node) against the reference's spelling, P/model/pretrain_goat.py:486-499,519-534."""
    torch.manual_seed(11)
    B, H, tau = 12, 768, 0.07
    mk = lambda: torch.nn.functional.normalize(torch.randn(B, H, device=DEV), dim=1).requires_grad_(True)
    go, vo, to = mk(), mk(), mk()
    fwl = torch.randn(B, 1, device=DEV).to(dtype).requires_grad_(True)

    def ref_chain():
        fw = torch.sigmoid(fwl.float())
        fo = go * fw + vo * (1 - fw)
        tgt = torch.arange(B, device=DEV)
        sym = lambda x: (torch.nn.functional.cross_entropy(x @ to.T / tau, tgt, reduction='none')
                         + torch.nn.functional.cross_entropy(to @ x.T / tau, tgt, reduction='none')) / 2
        return fo, sym(go) + sym(vo) + sym(fo)
    fo_ref, loss_ref = ref_chain()
    w = torch.randn(B, device=DEV)
    (loss_ref * w).sum().backward()
    g_ref = [t.grad.clone() for t in (go, vo, fwl, to)]
    for t in (go, vo, fwl, to):
        t.grad = None
    fo = ops.cfp_mix(go, vo, fwl)
    _close(fo, fo_ref, torch.float32, 'fo')
    loss = ops.cfp_tail(go, vo, fwl, to, tau)
    _close(loss, loss_ref, torch.float32, 'loss')
    (loss * w).sum().backward()
    for name, t, r in zip(('dgo', 'dvo', 'dfwl', 'dto'), (go, vo, fwl, to), g_ref):
        _close(t.grad, r, dtype if name == 'dfwl' else torch.float32, name)
    # the mix alone, with its own backward
    for t in (go, vo, fwl):
        t.grad = None
    gsel = torch.randn(B, H, device=DEV)
    (ops.cfp_mix(go, vo, fwl) * gsel).sum().backward()
    got = [t.grad.clone() for t in (go, vo, fwl)]
    for t in (go, vo, fwl):
        t.grad = None
    fw = torch.sigmoid(fwl.float())
    ((go * fw + vo * (1 - fw)) * gsel).sum().backward()
    for name, a, t in zip(('dgo', 'dvo', 'dfwl'), got, (go, vo, fwl)):
        _close(a, t.grad, dtype if name == 'dfwl' else torch.float32, 'mix ' + name)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(48, 22, 768), (48, 768), (1776, 768), (5, 256)])
def test_linear_with_one_output_column_through_the_rowdot_kernels(ops, dtype, shape):
    """Linear(H, 1) (last layer of ClsPrediction): goat_rowdot_fwd/_bwd == F.linear on the same (dtype-rounded) operands, gradients of x, W, b."""
    torch.manual_seed(5)
    H = shape[-1]
    w = (torch.randn(1, H, device=DEV) * 0.05).requires_grad_(True)
    b = torch.randn(1, device=DEV).requires_grad_(True)
    x = torch.randn(*shape, device=DEV).to(dtype).requires_grad_(True)
    y = ops.linear(x, w, b)
    assert y.shape == shape[:-1] + (1,) and y.dtype == dtype
    g = torch.randn_like(y)
    y.backward(g)
    got = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    wq = w.to(dtype).float()
    ref = torch.nn.functional.linear(x.float(), wq, b)
    ref.backward(g.float())
    _close(got[0], ref, dtype, 'y')
    _close(got[1], x.grad, dtype, 'dx')
    _close(got[2], w.grad, dtype, 'dW')
    _close(got[3], b.grad, dtype, 'db')


@pytest.mark.parametrize('N', [64, 60, 37])
def test_cross_entropy_rows_with_masked_actions_and_ignored_rows(ops, N):
    """hipops.cross_entropy_rows (the navigation step's action loss): == F.cross_entropy(reduction='none', ignore_index=-100) on logits with
    -inf entries (masked actions) and ignored rows; gradient included.  Widths that are no multiple of 4 take the torch path."""
    torch.manual_seed(N)
    B = 12
    lg = torch.randn(B, N, device=DEV) * 3
    lg[:, N - 5:] = float('-inf')
    lg[3, 1:7] = float('-inf')
    lg.requires_grad_(True)
    tg = torch.randint(0, N - 5, (B,), device=DEV)
    tg[3] = 0
    tg[5] = -100
    w = torch.randn(B, device=DEV)
    got = ops.cross_entropy_rows(lg, tg)
    (got * w).sum().backward()
    g_got = lg.grad.clone()
    lg.grad = None
    ref = torch.nn.functional.cross_entropy(lg, tg, reduction='none', ignore_index=-100)
    (ref * w).sum().backward()
    assert float(got[5]) == 0.0
    _close(got, ref, torch.float32, 'loss')
    assert torch.equal(torch.isfinite(g_got), torch.isfinite(lg.grad))
    _close(torch.nan_to_num(g_got), torch.nan_to_num(lg.grad), torch.float32, 'dlogits')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,Lq,Lk,n', [(6, 23, 80, 6), (3, 80, 37, 3)])
def test_linear_bank_matches_one_projection_per_layer(ops, dtype, B, Lq, Lk, n):
    """All layers' cross-attention K|V projections of one attended sequence as ONE GEMM (hipops.linear_bank; reference loop
    P/model/Bert_backbone.py:765-781) against one multi_linear per layer + fan-out: forward results identical bit for bit; the
    gradients agree within the dtype's tolerance with a float64 reference (the bank rounds the gradient of the attended sequence
    ONCE, the per-layer path n + 1 times) — and attention writes its key|value gradient straight into the bank's buffer."""
    H, nh = 768, 12
    g = torch.Generator().manual_seed(3)
    x0 = (torch.randn(B, Lk, H, generator=g) * 0.5)
    q0 = [(torch.randn(B, Lq, H, generator=g) * 0.5) for _ in range(n)]
    Ws = [[torch.randn(H, H, generator=g) * 0.03 for _ in range(2)] for _ in range(n)]
    bs = [[torch.randn(H, generator=g) * 0.1 for _ in range(2)] for _ in range(n)]
    do = [torch.randn(B, Lq, H, generator=g) for _ in range(n)]
    kmask = torch.zeros(B, Lk)
    kmask[0, Lk - 5:] = -10000.0

    def run(bank):
        x = x0.to(DEV, dtype).requires_grad_()
        qs = [t.to(DEV, dtype).requires_grad_() for t in q0]
        params = [[(torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))) for w, b in zip(Wl, bl)] for Wl, bl in zip(Ws, bs)]
        if bank:
            kvs = ops.linear_bank(x, params)
            assert all(kv.stride(1) == n * 2 * H for kv in kvs)
        else:
            kvs = [ops.multi_linear(h, [p[0] for p in grp], [p[1] for p in grp]) for h, grp in zip(ops.fanout(x, n), params)]
        outs = [ops.attention(qs[i], kvs[i], kmask.to(DEV), None, nh, 0.0) for i in range(n)]
        loss = sum((o.float() * d.to(DEV)).sum() for o, d in zip(outs, do))
        loss.backward()
        torch.cuda.synchronize()
        return ([kv.detach().float().cpu() for kv in kvs], [o.detach().float().cpu() for o in outs], x.grad.float().cpu(),
                [t.grad.float().cpu() for t in qs], [[(w.grad.cpu(), b.grad.cpu()) for w, b in grp] for grp in params])

    kv_b, o_b, dx_b, dq_b, dp_b = run(True)
    kv_l, o_l, dx_l, dq_l, dp_l = run(False)
    for a, b in zip(kv_b + o_b, kv_l + o_l):
        assert torch.equal(a, b), 'the bank changes a forward result'
    for a, b in zip(dq_b, dq_l):
        assert torch.equal(a, b)
    # float64 reference of the same graph
    x = x0.double().requires_grad_()
    qs = [t.double().requires_grad_() for t in q0]
    if dtype == torch.bfloat16:
        x = x0.bfloat16().double().requires_grad_()
        qs = [t.bfloat16().double().requires_grad_() for t in q0]
    pr = [[(w.double().requires_grad_() if dtype == torch.float32 else w.bfloat16().double().requires_grad_(), b.double().requires_grad_())
           for w, b in zip(Wl, bl)] for Wl, bl in zip(Ws, bs)]
    loss = 0
    for i in range(n):
        k = (x @ pr[i][0][0].t() + pr[i][0][1]).view(B, Lk, nh, 64).transpose(1, 2)
        v = (x @ pr[i][1][0].t() + pr[i][1][1]).view(B, Lk, nh, 64).transpose(1, 2)
        qh = qs[i].view(B, Lq, nh, 64).transpose(1, 2)
        s = qh @ k.transpose(-1, -2) / 8.0 + kmask.double()[:, None, None, :]
        o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, Lq, H)
        loss = loss + (o * do[i].double()).sum()
    loss.backward()
    _close(dx_b, x.grad, dtype, 'bank dx')
    _close(dx_l, x.grad, dtype, 'per-layer dx')
    e_b = (dx_b.double() - x.grad).abs().max()
    e_l = (dx_l.double() - x.grad).abs().max()
    if dtype == torch.bfloat16:
        assert e_b <= 1.5 * e_l + 1e-6, (float(e_b), float(e_l))    # one rounding instead of n + 1
    for i in range(n):
        for j in range(2):
            _close(dp_b[i][j][0], pr[i][j][0].grad, dtype, 'bank dW %d %d' % (i, j))
            if j == 1:
                _close(dp_b[i][j][1], pr[i][j][1].grad, dtype, 'bank db %d %d' % (i, j))
            else:       # the key bias shifts every score of a query row alike: softmax is invariant, its gradient is zero up to rounding
                assert float(dp_b[i][0][1].abs().max()) <= _tol(dtype) * float(pr[i][1][1].grad.abs().max()), 'bank db (key) %d' % i
            _close(dp_b[i][j][0], dp_l[i][j][0], dtype, 'dW vs per-layer')


@pytest.mark.parametrize('M', [515, 3840, 8641])
@pytest.mark.parametrize('form', ['plain', 'res_drop_fork', 'z_out', 'fork_in', 'out_drop_post'])
def test_layer_norm_backward_768_kernel_matches_the_generic_kernel(ops, M, form):
    """ln_bwd768_kernel (bf16 rows of 768: 8-byte chunks, packed rows in flight, <= 128 VGPRs) against the generic ln_bwd_kernel on the same
    inputs and dropout counters: same column partials, row statistics summed in another lane order — equal to float rounding."""
    H = 768
    g = torch.Generator().manual_seed(M)
    x0 = torch.randn(M, H, generator=g)
    r0 = torch.randn(M, H, generator=g)
    dy0, dy1 = torch.randn(M, H, generator=g), torch.randn(M, H, generator=g)
    gamma0, beta0 = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)

    def run(generic):
        if generic:
            os.environ['GOAT_LN_BWD_GENERIC'] = '1'
        else:
            os.environ.pop('GOAT_LN_BWD_GENERIC', None)
        try:
            ops.manual_seed(11)
            x = x0.to(DEV, torch.bfloat16).requires_grad_()
            r = r0.to(DEV, torch.bfloat16).requires_grad_()
            gamma, beta = gamma0.to(DEV).requires_grad_(), beta0.to(DEV).requires_grad_()
            a, b = dy0.to(DEV, torch.bfloat16), dy1.to(DEV, torch.bfloat16)
            if form == 'plain':
                y = ops.layer_norm(x, gamma, beta, 1e-12)
                y.backward(a)
            elif form == 'res_drop_fork':
                y1, y2 = ops.layer_norm(x, gamma, beta, 1e-12, r, 0.1, fork=True)
                torch.autograd.backward([y1, y2], [a, b])
            elif form == 'z_out':
                y, zz = ops.layer_norm(x, gamma, beta, 1e-5, r, 0.1, z_out=True)
                torch.autograd.backward([y, zz], [a, b])
            elif form == 'fork_in':
                y, xs = ops.layer_norm(x, gamma, beta, 1e-5, fork_in=True)
                torch.autograd.backward([y, xs], [a, b])
            else:
                post = r
                y = ops.layer_norm(x, gamma, beta, 1e-12, p_out=0.1, post_add=post)
                y.backward(a)
            torch.cuda.synchronize()
            return [t.grad.float().cpu() for t in (x, r, gamma, beta) if t.grad is not None]
        finally:
            os.environ.pop('GOAT_LN_BWD_GENERIC', None)

    new, old = run(False), run(True)
    assert len(new) == len(old) and len(new) >= 3
    for i, (a, b) in enumerate(zip(new, old)):
        scale = b.abs().max().clamp_min(1e-6)
        assert (a - b).abs().max() / scale < 1e-2, (form, i, float((a - b).abs().max() / scale))       # (a bf16 ulp where a rounding flips)
        assert (a - b).abs().mean() / scale < 2e-5, (form, i, float((a - b).abs().mean() / scale))
