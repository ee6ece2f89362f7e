"""Host side of the fresh-batch step (CPU): the vectorised index builders against per-token loop restatements, and
train_step.StaticBatch (pack / stage / commit of a new host batch into the fixed-address batch + mask refresh)."""
import numpy as np
import pytest
import torch

from helpers import ROOT  # noqa: F401  (puts the repo root on sys.path)


def _vp_loop(step_lens, view_lens, V):
    last = np.cumsum(step_lens) - 1
    vp_lens = [int(view_lens[n]) + 1 for n in last]
    width = max(vp_lens)
    idx, start = [], [0]
    for b in range(len(step_lens)):
        for j in range(width):
            if j >= 1:
                idx.append(int(last[b]) * V + j - 1)
            start.append(len(idx))
    return idx, start, vp_lens, width


def _objcat_loop(vl, ol, V, O, W):
    N = len(vl)
    idx, start = [], [0]
    for n in range(N):
        for j in range(W):
            if j < vl[n]:
                idx.append(n * V + j)
            elif j < vl[n] + ol[n]:
                idx.append(N * V + n * O + j - vl[n])
            start.append(len(idx))
    return idx or [-1], start


def test_vectorised_index_builders_match_loops():
    from vln_goat_amd import graphmap
    rs = np.random.RandomState(3)
    for _ in range(60):
        B = rs.randint(1, 6)
        step = rs.randint(1, 5, B).tolist()
        N, V = sum(step), rs.randint(3, 9)
        vl = rs.randint(1, V + 1, N)
        idx, start, vp_lens, width = graphmap.build_vp_index(step, torch.tensor(vl), V)
        ridx, rstart, rlens, rwidth = _vp_loop(step, vl, V)
        assert idx.tolist() == ridx and start.tolist() == rstart and vp_lens.tolist() == rlens and width == rwidth
        assert idx.dtype == torch.int32 and start.dtype == torch.int32 and vp_lens.dtype == torch.int64
        O = rs.randint(1, 5)
        ol = rs.randint(0, O + 1, N)
        W = int((vl + ol).max()) + rs.randint(0, 2)
        idx, start = graphmap.build_obj_concat_index(torch.tensor(vl), torch.tensor(ol), V, O, W)
        ridx, rstart = _objcat_loop(vl, ol, V, O, W)
        assert idx.tolist() == ridx and start.tolist() == rstart
    with pytest.raises(ValueError):
        graphmap.build_obj_concat_index([4], [3], 4, 3, 6)


def test_inverse_gather_index_matches_brute_force():
    from vln_goat_amd import graphmap
    rs = np.random.RandomState(11)
    for trial in range(40):
        n_src, n_seg = rs.randint(1, 30), rs.randint(1, 12)
        segs = [rs.randint(0, n_src, rs.randint(0, 5)).tolist() for _ in range(n_seg)]
        flat = [i for s in segs for i in s]
        idx = torch.tensor(flat or [-1], dtype=torch.int32)
        start = torch.tensor(np.cumsum([0] + [len(s) for s in segs]), dtype=torch.int32)
        scale = torch.tensor(rs.rand(n_seg).astype(np.float32))
        inv_idx, inv_start, inv_w = graphmap.inverse_index(idx, start, scale, n_src)
        assert inv_start.shape[0] == n_src + 1 and int(inv_start[-1]) == len(flat)
        for r in range(n_src):
            want = sorted((k, float(scale[k])) for k, s in enumerate(segs) for i in s if i == r)
            a, b = int(inv_start[r]), int(inv_start[r + 1])
            got = sorted((int(inv_idx[j]), float(inv_w[j])) for j in range(a, b))
            assert got == want, (trial, r)
        assert graphmap.inverse_index(idx, start, None, n_src)[2] is None
    with pytest.raises(ValueError):
        graphmap.inverse_index(torch.tensor([5], dtype=torch.int32), torch.tensor([0, 1], dtype=torch.int32), None, 3)


def test_static_batch_pack_commit_and_shape_guard():
    from vln_goat_amd import config as gcfg, layers, synth, train_step
    cfg = gcfg.make_config(num_l_layers=1, num_top_layer=1, num_pano_layers=1, vocab_size=300)
    mk = lambda seed, **kw: synth.make_pretrain_batch(**{**dict(B=5, T=3, L=24, seed=seed, vocab_size=300, style='survey'), **kw})
    first, second = mk(1), mk(2)
    sb = train_step.StaticBatch(cfg, first, device='cpu')
    addr = {k: v.data_ptr() for k, v in sb.gb.items() if torch.is_tensor(v)}
    for k, v in first.items():
        if torch.is_tensor(v):
            assert torch.equal(sb.gb[k], v), k
    # a mask memoised on the static batch before the swap follows the new data afterwards, at the same address
    lens = sb.gb['txt_lens']
    lens.copy_(torch.tensor([24, 3, 7, 24, 1]))
    m = layers.gen_seq_masks(lens, 24)
    neg = layers.neg_mask(m)
    m_ptr, neg_ptr = m.data_ptr(), neg.data_ptr()
    second['txt_lens'] = torch.tensor([2, 24, 9, 4, 24])
    buf = sb.pack(second)
    sb.stage(buf)
    sb.commit()
    for k, v in second.items():
        if torch.is_tensor(v):
            assert torch.equal(sb.gb[k], v) and sb.gb[k].data_ptr() == addr[k], k
    idx = train_step.collate_indices(cfg, second)
    c = sb.gb['_goat_cache']
    for k in ('gmap', 'vp', 'sap', 'gmap_inv', 'vp_inv'):
        for a, b in zip(c[k], idx[k]):
            assert (torch.equal(a, b) if torch.is_tensor(a) else a == b), k
    assert torch.equal(c['mlm_idx'], idx['mlm_idx']) and torch.equal(c['mlm_tgt'], idx['mlm_tgt'])
    want = torch.arange(24)[None] < second['txt_lens'][:, None]
    assert m.data_ptr() == m_ptr and torch.equal(m, want)
    assert neg.data_ptr() == neg_ptr and torch.equal(neg, (1.0 - want.float()) * -10000.0)
    assert layers.gen_seq_masks(lens, 24) is m                       # still memoised (no recompute on the next call)
    # another shape is refused, never silently truncated
    with pytest.raises(ValueError):
        sb.pack(mk(3, T=2))
    with pytest.raises(ValueError):
        sb.pack(mk(3, L=20))
    with pytest.raises(RuntimeError):
        sb.commit()
