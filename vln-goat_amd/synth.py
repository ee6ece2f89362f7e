"""Synthetic GOAT pre-training batches and seeded weights (no dataset / checkpoint is reachable offline).

The batch follows the reference collate schema exactly (P/data/tasks.py mlm/sap/cfp_collate; field list in
SURVEY.md §8a-0): same keys, dtypes, python-list fields (viewpoint-id strings) and padding conventions.
Values come from numpy's legacy `RandomState` (bit-stable across numpy versions), so golden fixtures only
need to store seeds: tests/golden/make_golden_pretrain.py feeds these very tensors to the imported
reference.
"""
import numpy as np
import torch


def _angle_fts(rs, n):
    # P/data/dataset.py:755-760 — sin/cos heading, sin/cos elevation, box (1,1,1)
    h = rs.uniform(-np.pi, np.pi, n)
    e = rs.uniform(-0.5, 0.5, n)
    return np.stack([np.sin(h), np.cos(h), np.sin(e), np.cos(e), np.ones(n), np.ones(n), np.ones(n)], 1).astype(np.float32)


def make_pretrain_batch(B=4, T=5, L=80, seed=0, vocab_size=50265, n_views=36, n_cand=4, style='survey',
                        ragged_views=False, mask_prob=0.15, feat_dim=768):
    """One batch usable for all of mlm / sap / cfp.

    T, L: int (fixed) or list of per-sample values.  style='survey': every step sees `n_cand` fresh
    candidates, one of which is the next path node (G = 2 + 4T - ... = 22 at T=5).
    style='rich': additionally a back-edge to the previous node and an unvisited node shared between
    consecutive steps (exercises the visited-candidate and multi-view-mean branches of the reference loops).
    """
    rs = np.random.RandomState(seed)
    Ts = [T] * B if isinstance(T, int) else list(T)
    Ls = [L] * B if isinstance(L, int) else list(L)
    Lmax = max(Ls)
    N = sum(Ts)

    txt_ids = np.zeros((B, Lmax), dtype=np.int64)
    txt_labels = -np.ones((B, Lmax), dtype=np.int64)
    for b in range(B):
        txt_ids[b, :Ls[b]] = rs.randint(3, vocab_size, Ls[b])
        nmask = max(1, int(round(mask_prob * Ls[b])))
        pos = rs.choice(Ls[b], nmask, replace=False)
        txt_labels[b, pos] = rs.randint(3, vocab_size, nmask)

    view_lens = rs.randint(n_views - 6, n_views + 1, N) if ragged_views else np.full(N, n_views)
    V = int(view_lens.max())
    fts = rs.standard_normal((N, V, feat_dim)).astype(np.float32)
    loc = np.stack([_angle_fts(rs, V) for _ in range(N)], 0)
    nav_types = np.zeros((N, V), dtype=np.int64)
    for n in range(N):
        fts[n, view_lens[n]:] = 0
        loc[n, view_lens[n]:] = 0

    traj_vpids, traj_cand_vpids, gmap_vpids, gmap_step_ids, gmap_visited = [], [], [], [], []
    n = 0
    for b in range(B):
        path = ['s%d_p%d' % (b, t) for t in range(Ts[b])]
        cands_all = []
        visited, unvisited = {}, {}
        shared_prev = None
        for t in range(Ts[b]):
            cands = []
            if t + 1 < Ts[b]:
                cands.append(path[t + 1])
            if style == 'rich':
                if t > 0:
                    cands.append(path[t - 1])
                if shared_prev is not None:
                    cands.append(shared_prev)
            k = 0
            while len(cands) < n_cand:
                cands.append('s%d_u%d_%d' % (b, t, k))
                k += 1
            order = rs.permutation(len(cands))
            cands = [cands[i] for i in order]
            shared_prev = [c for c in cands if '_u' in c][-1] if style == 'rich' else None
            cands_all.append(cands)
            nav_types[n + t, :len(cands)] = 1
            # same bookkeeping as P/data/dataset.py:515-522
            visited[path[t]] = t + 1
            unvisited.pop(path[t], None)
            for c in cands:
                if c not in visited:
                    unvisited[c] = 0
        n += Ts[b]
        traj_vpids.append(path)
        traj_cand_vpids.append(cands_all)
        gmap_vpids.append([None] + list(visited.keys()) + list(unvisited.keys()))
        gmap_step_ids.append([0] + list(visited.values()) + list(unvisited.values()))
        gmap_visited.append([0] + [1] * len(visited) + [0] * len(unvisited))

    gmap_lens = np.array([len(x) for x in gmap_vpids], dtype=np.int64)
    G = int(gmap_lens.max())
    step_ids = np.zeros((B, G), dtype=np.int64)
    vis = np.zeros((B, G), dtype=bool)
    pos_fts = np.zeros((B, G, 7), dtype=np.float32)
    pair = np.zeros((B, G, G), dtype=np.float32)
    global_lab = np.zeros(B, dtype=np.int64)
    local_lab = np.zeros(B, dtype=np.int64)
    for b in range(B):
        g = gmap_lens[b]
        step_ids[b, :g] = gmap_step_ids[b]
        vis[b, :g] = gmap_visited[b]
        pos_fts[b, :g] = rs.standard_normal((g, 7)).astype(np.float32)
        d = rs.uniform(0, 1, (g, g)).astype(np.float32)
        d = np.triu(d, 1)
        d = d + d.T
        d[0, :] = 0
        d[:, 0] = 0
        pair[b, :g, :g] = d
        # labels: a valid (unvisited or stop) global slot; the matching local candidate when there is one
        last_c = traj_cand_vpids[b][-1]
        choices = [0] + [i for i in range(1, g) if not gmap_visited[b][i]]
        gi = int(choices[rs.randint(len(choices))])
        global_lab[b] = gi
        vp = gmap_vpids[b][gi]
        local_lab[b] = (last_c.index(vp) + 1) if (gi > 0 and vp in last_c) else 0

    last = np.cumsum(Ts) - 1
    vp_w = int(view_lens[last].max()) + 1
    vp_pos = rs.standard_normal((B, vp_w, 14)).astype(np.float32)
    for b in range(B):
        vp_pos[b, view_lens[last[b]] + 1:] = 0

    t = torch.from_numpy
    return {
        'txt_ids': t(txt_ids), 'txt_lens': torch.tensor(Ls, dtype=torch.int64), 'txt_labels': t(txt_labels),
        'traj_view_img_fts': t(fts), 'traj_loc_fts': t(loc), 'traj_nav_types': t(nav_types),
        'traj_step_lens': list(Ts), 'traj_vp_view_lens': t(view_lens.astype(np.int64)),
        'traj_vpids': traj_vpids, 'traj_cand_vpids': traj_cand_vpids, 'gmap_vpids': gmap_vpids,
        'gmap_lens': t(gmap_lens), 'gmap_step_ids': t(step_ids), 'gmap_pos_fts': t(pos_fts),
        'gmap_pair_dists': t(pair), 'gmap_visited_masks': t(vis), 'vp_pos_fts': t(vp_pos),
        'global_act_labels': t(global_lab), 'local_act_labels': t(local_lab),
        'extra_heads': [True] * B, 'traj_reverie_loc_fts': None,
    }


def batch_to(batch, device):
    """PrefetchLoader-style host->device move (P/data/loader.py:109-115): tensors only, lists stay."""
    out = {}
    for k, v in batch.items():
        out[k] = v.to(device, non_blocking=True) if torch.is_tensor(v) else v
    return out


def n_traj_steps(batch):
    return int(sum(batch['traj_step_lens']))


def seeded_state_dict(model, seed=0, perturb=True):
    """Deterministic weights for every entry of `model.state_dict()` (numpy RandomState keyed by the
    parameter name): N(0, 0.02) matrices / embeddings, LayerNorm (1, 0), zero biases, U(-0.1, 0.1)
    `tim_*_attn`.  perturb=True additionally randomises biases and LayerNorm affine parameters so that
    parity tests exercise them.  Tied tensors (MLM decoder = word embeddings) receive identical values."""
    import zlib
    sd = model.state_dict()
    ln_names = set()
    for mname, m in model.named_modules():
        if isinstance(m, torch.nn.LayerNorm):
            ln_names.add(mname)
    out = {}
    for name, ref in sd.items():
        if name.endswith('position_ids') or name.endswith('token_type_ids'):
            out[name] = ref.clone()
            continue
        key = 'bert.embeddings.word_embeddings.weight' if name == 'mlm_head.predictions.decoder.weight' else name
        rs = np.random.RandomState((zlib.crc32(key.encode()) + seed * 1000003) % (2 ** 31))
        shape = tuple(ref.shape)
        mod = name.rsplit('.', 1)[0]
        leaf = name.rsplit('.', 1)[-1]
        if mod in ln_names:
            if leaf == 'weight':
                v = 1.0 + (0.1 * rs.standard_normal(shape) if perturb else 0.0) * np.ones(shape)
            else:
                v = (0.05 * rs.standard_normal(shape)) if perturb else np.zeros(shape)
        elif name.startswith('tim_') and name.endswith('_attn'):
            v = rs.uniform(-0.1, 0.1, shape)
        elif leaf in ('bias', 'in_proj_bias'):
            v = (0.02 * rs.standard_normal(shape)) if perturb else np.zeros(shape)
        elif name.endswith('sprel_linear.weight'):
            v = 0.5 + 0.1 * rs.standard_normal(shape)
        else:
            v = 0.02 * rs.standard_normal(shape)
        out[name] = torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape).to(ref.dtype)
    return out
