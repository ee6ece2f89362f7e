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
    # (the 7- / 14-wide position features are stored in the form their Linear consumes: compute dtype, K zero-padded to a 16-byte chunk)
    prep = train_step.prepare_position_features
    assert prep(first)['traj_loc_fts'].shape[-1] == 8 and prep(first)['vp_pos_fts'].shape[-1] == 16
    assert torch.equal(prep(first)['gmap_pos_fts'][..., :7], first['gmap_pos_fts']) and not prep(first)['gmap_pos_fts'][..., 7:].any()
    for k, v in prep(first).items():
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
    for k, v in prep(second).items():
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


def _eval_csr(src, idx, start, scale, n_out):
    idx, start = idx.numpy(), start.numpy()
    out = np.zeros((n_out, src.shape[1]), np.float64)
    for s in range(n_out):
        seg = idx[start[s]:start[s + 1]]
        if len(seg):
            out[s] = src[seg].sum(0) * (1.0 if scale is None else float(scale[s]))
    return out


def test_shape_bucket_padding_keeps_every_real_token():
    """pad_batch + capacity-padded indices (train_step.StaticBatch(bucket=...)): a RAGGED batch padded into a bucket yields, for
    every real output slot, the same gathered rows as its own unpadded indices (evaluated with numpy on random source rows) —
    map tokens (fused panorama rows moved behind the padded view rows), local tokens, inverse indices, MLM selection, SAP
    fusion — and only ignorable padding elsewhere."""
    from vln_goat_amd import config as gcfg, graphmap, synth, train_step
    cfg = gcfg.make_config(num_l_layers=1, num_top_layer=1, num_pano_layers=1, vocab_size=300)
    rs = np.random.RandomState(0)
    for trial in range(6):
        B = 4
        T = rs.randint(1, 5, B).tolist()
        L = rs.randint(8, 25, B).tolist()
        batch = synth.make_pretrain_batch(B=B, T=T, L=L, seed=trial, vocab_size=300, style='rich')
        own = train_step.collate_indices(cfg, batch)
        N, V = batch['traj_view_img_fts'].shape[:2]
        G, W = batch['gmap_step_ids'].shape[1], batch['vp_pos_fts'].shape[1]
        bucket = dict(L=32, N=N + rs.randint(0, 5), G=G + rs.randint(0, 4), W=W)
        padded = train_step.pad_batch(batch, **bucket)
        assert padded['txt_ids'].shape == (B, 32) and padded['traj_view_img_fts'].shape[0] == bucket['N']
        assert int((padded['txt_labels'] != -1).sum()) == int((batch['txt_labels'] != -1).sum())
        caps = train_step.index_capacities(padded, ('mlm', 'sap', 'cfp'))
        idx = train_step.collate_indices(cfg, padded, caps=caps, vp_width=W)
        H = 5
        Np = bucket['N']
        views, fused = rs.standard_normal((N, V, H)), rs.standard_normal((N, H))
        src_own = np.concatenate([views.reshape(N * V, H), fused], 0)
        pv = np.concatenate([views, rs.standard_normal((Np - N, V, H))], 0)
        pf = np.concatenate([fused, rs.standard_normal((Np - N, H))], 0)
        src_pad = np.concatenate([pv.reshape(Np * V, H), pf], 0)
        a = _eval_csr(src_own, own['gmap'][0], own['gmap'][1], own['gmap'][2], B * G).reshape(B, G, H)
        b = _eval_csr(src_pad, idx['gmap'][0], idx['gmap'][1], idx['gmap'][2], B * bucket['G']).reshape(B, bucket['G'], H)
        assert np.allclose(a, b[:, :G]) and not b[:, G:].any()
        a = _eval_csr(views.reshape(N * V, H), own['vp'][0], own['vp'][1], None, B * W).reshape(B, W, H)
        b = _eval_csr(pv.reshape(Np * V, H), idx['vp'][0], idx['vp'][1], None, B * W).reshape(B, W, H)
        assert np.allclose(a, b) and idx['vp'][3] == W
        assert idx['gmap'][0].shape[0] == caps['nnz_gmap'] and idx['gmap_inv'][0].shape[0] == caps['nnz_gmap']
        # inverse index of the padded gather: source row r is read by exactly the segments that list it
        inv_idx, inv_start, inv_w = idx['gmap_inv']
        gi, gs = idx['gmap'][0].numpy(), idx['gmap'][1].numpy()
        for r in rs.randint(0, src_pad.shape[0], 12):
            want = sorted(s for s in range(len(gs) - 1) if r in gi[gs[s]:gs[s + 1]])
            assert sorted(inv_idx[int(inv_start[r]):int(inv_start[r + 1])].tolist()) == want
        # MLM: the real rows first, padding rows ignored; the scale restores the mean
        n = int(own['mlm_idx'].shape[0])
        rows_own = own['mlm_idx'].numpy()
        b_own, pos_own = rows_own // batch['txt_ids'].shape[1], rows_own % batch['txt_ids'].shape[1]
        rows_pad = idx['mlm_idx'].numpy()[:n]
        assert np.array_equal(rows_pad // 32, b_own) and np.array_equal(rows_pad % 32, pos_own)
        assert torch.equal(idx['mlm_tgt'][:n], own['mlm_tgt']) and bool((idx['mlm_tgt'][n:] == -100).all())
        assert abs(float(idx['mlm_scale']) - caps['mlm'] / n) < 1e-6
        assert torch.equal(idx['sap'][1][:, :G], own['sap'][1]) and not bool(idx['sap'][1][:, G:].any())
    with pytest.raises(ValueError):
        train_step.pad_batch(batch, L=4)


def test_static_batch_bucket_accepts_ragged_batches():
    from vln_goat_amd import config as gcfg, synth, train_step
    cfg = gcfg.make_config(num_l_layers=1, num_top_layer=1, num_pano_layers=1, vocab_size=300)
    mk = lambda seed, T, L: synth.make_pretrain_batch(B=4, T=T, L=L, seed=seed, vocab_size=300, style='survey')
    first = mk(1, [3, 2, 4, 3], [20, 24, 9, 16])
    bucket = dict(L=24, N=16, G=max(first['gmap_step_ids'].shape[1], 18))
    sb = train_step.StaticBatch(cfg, first, device='cpu', bucket=bucket)
    assert sb.gb['txt_ids'].shape == (4, 24) and sb.gb['traj_view_img_fts'].shape[0] == 16
    other = mk(2, [1, 4, 2, 2], [24, 5, 12, 7])
    assert sb.fits(other)
    sb.stage(sb.pack(other))
    sb.commit()
    n = other['traj_view_img_fts'].shape[0]
    assert torch.equal(sb.gb['traj_view_img_fts'][:n], other['traj_view_img_fts'])
    assert torch.equal(sb.gb['txt_ids'][:, :other['txt_ids'].shape[1]], other['txt_ids'])
    assert torch.equal(sb.gb['gmap_lens'], other['gmap_lens'])
    too_long = mk(3, [5, 5, 5, 5], [24, 24, 24, 24])
    assert not sb.fits(too_long)
    with pytest.raises(ValueError):
        sb.pack(too_long)


def test_collate_indices_covers_object_batches():
    """REVERIE / SOON batches (object tokens, OG and MRC heads): every index the model otherwise builds lazily on the device side
    (pretrain_model._indices / forward_og / _mrc_rows) comes out of collate_indices on the host — compared with the model's own
    lazily built cache on the same batch."""
    from vln_goat_amd import config as gcfg, pretrain_model, synth, train_step
    cfg = gcfg.make_config(num_l_layers=1, num_top_layer=1, num_pano_layers=1, vocab_size=300, dataset='reverie', obj_feat_size=768,
                           pretrain_tasks=['mlm', 'mrc', 'sap', 'og', 'cfp'])
    batch = synth.make_pretrain_batch(B=3, T=[2, 3, 1], L=[12, 9, 14], seed=4, vocab_size=300, style='rich', objects=4, mrc=True)
    idx = train_step.collate_indices(cfg, batch, tasks=('mlm', 'mrc', 'sap', 'og', 'cfp'))
    model = pretrain_model.GlocalTextPathCMTPreTraining(cfg)
    b2 = dict(batch)
    cache = model.bert._indices(b2)
    for k in ('objcat', 'objcat_inv', 'gmap', 'gmap_inv', 'vp_inv'):
        for a, b in zip(cache[k], idx[k]):
            assert torch.equal(a, b) if torch.is_tensor(a) else a == b, k
    for a, b in zip(cache['vp'], idx['vp']):
        assert torch.equal(a, b) if torch.is_tensor(a) else a == b
    vl, ol = model._last_lens(b2)
    W1 = idx['vp'][3]
    for which in ('view', 'obj'):
        rows, sel = model._mrc_rows(b2, which, W1)
        assert torch.equal(rows.cpu(), idx['mrc_' + which][0]) and torch.equal(sel.cpu(), idx['mrc_' + which][1])
    oi, om = idx['og_idx']
    for b, (v, o) in enumerate(zip(vl, ol)):
        assert oi[b, :o].tolist() == list(range(1 + v, 1 + v + o)) and om[b].tolist() == [True] * o + [False] * (om.shape[1] - o)


def test_collate_refuses_action_labels_outside_their_logit_row():
    """goat_sap_fuse treats a label outside its row like the ignore value (loss 0, no gradient); F.cross_entropy, which it replaces, raises.
    collate_indices therefore validates the labels on the host."""
    from vln_goat_amd import config as gcfg, synth, train_step
    cfg = gcfg.make_config(num_l_layers=1, num_top_layer=1, num_pano_layers=1, vocab_size=300)
    b = synth.make_pretrain_batch(B=3, T=2, L=12, seed=4, vocab_size=300, style='survey')
    train_step.collate_indices(cfg, b, ('sap',))                      # valid labels pass
    G, W = b['gmap_step_ids'].shape[1], b['vp_pos_fts'].shape[1]
    ok = dict(b)
    ok['global_act_labels'] = b['global_act_labels'].clone()
    ok['global_act_labels'][0] = -100                                  # the ignore value stays legal
    train_step.collate_indices(cfg, ok, ('sap',))
    for key, width in (('global_act_labels', G), ('local_act_labels', W)):
        bad = dict(b)
        bad[key] = b[key].clone()
        bad[key][1] = width
        with pytest.raises(ValueError):
            train_step.collate_indices(cfg, bad, ('sap',))
        bad[key][1] = -3
        with pytest.raises(ValueError):
            train_step.collate_indices(cfg, bad, ('sap',))


def test_position_features_are_prepared_once_for_every_dtype():
    from vln_goat_amd import synth, train_step
    b = synth.make_pretrain_batch(B=2, T=2, L=10, seed=2, vocab_size=300, style='survey')
    for dt, e in ((torch.bfloat16, 8), (torch.float32, 4)):
        p = train_step.prepare_position_features(b, dt)
        for k in ('traj_loc_fts', 'gmap_pos_fts', 'vp_pos_fts'):
            assert p[k].dtype == dt and p[k].shape[-1] % e == 0 and p[k].shape[:-1] == b[k].shape[:-1]
            n = b[k].shape[-1]
            assert torch.equal(p[k][..., :n], b[k].to(dt)) and not p[k][..., n:].any()
        assert p['txt_ids'] is b['txt_ids']                            # everything else untouched
        again = train_step.prepare_position_features(p, dt)            # idempotent
        assert all(again[k] is p[k] for k in ('traj_loc_fts', 'gmap_pos_fts', 'vp_pos_fts'))
