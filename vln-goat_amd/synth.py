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
                        ragged_views=False, mask_prob=0.15, feat_dim=768, objects=0, obj_dim=768, mrc=False,
                        prob_size=1000, zdict=None):
    """One batch usable for all of mlm / sap / cfp.

    T, L: int (fixed) or list of per-sample values.  style='survey': every step sees `n_cand` fresh
    candidates, one of which is the next path node (G = 2 + 4T - ... = 22 at T=5).
    style='rich': additionally a back-edge to the previous node and an unvisited node shared between
    consecutive steps (exercises the visited-candidate and multi-view-mean branches of the reference loops).
    objects > 0: REVERIE/SOON-style batch (P/data/tasks.py og_collate): up to `objects` object tokens per panorama
    (0 allowed except on a sample's last step), loc/nav-type tensors [N, max(view+obj)], nav type 2 = object,
    `obj_labels`.  mrc=True adds the masked-region inputs (soft labels over `prob_size` classes).
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
    extra = {}
    tot_lens = view_lens.copy()
    if objects > 0:
        obj_lens = rs.randint(0, objects + 1, N)
        obj_lens[last] = np.maximum(obj_lens[last], 1)
        O = int(obj_lens.max())
        obj_fts = rs.standard_normal((N, O, obj_dim)).astype(np.float32)
        obj_names = rs.randint(0, 45, (N, O)).astype(np.int64)
        tot_lens = view_lens + obj_lens
        W = int(tot_lens.max())
        loc2 = np.zeros((N, W, 7), dtype=np.float32)
        nav2 = np.zeros((N, W), dtype=np.int64)
        for n_ in range(N):
            obj_fts[n_, obj_lens[n_]:] = 0
            obj_names[n_, obj_lens[n_]:] = 0
            loc2[n_, :view_lens[n_]] = loc[n_, :view_lens[n_]]
            loc2[n_, view_lens[n_]:tot_lens[n_]] = _angle_fts(rs, int(obj_lens[n_]))
            nav2[n_, :view_lens[n_]] = nav_types[n_, :view_lens[n_]]
            nav2[n_, view_lens[n_]:tot_lens[n_]] = 2
        loc, nav_types = loc2, nav2
        extra.update({'traj_obj_img_fts': torch.from_numpy(obj_fts), 'traj_vp_obj_lens': torch.from_numpy(obj_lens.astype(np.int64)),
                      'traj_reverie_obj_names': torch.from_numpy(obj_names),
                      'obj_labels': torch.from_numpy(np.array([rs.randint(obj_lens[n_]) for n_ in last], dtype=np.int64))})
    vp_w = int(tot_lens[last].max()) + 1
    vp_pos = rs.standard_normal((B, vp_w, 14)).astype(np.float32)
    for b in range(B):
        vp_pos[b, tot_lens[last[b]] + 1:] = 0
    if mrc:
        def soft(n_rows):
            p = rs.uniform(0, 1, (n_rows, prob_size)).astype(np.float32) ** 4
            return p / p.sum(1, keepdims=True)
        vmax = int(view_lens[last].max())
        vm = np.zeros((B, vmax), dtype=bool)
        vprob = np.zeros((B, vmax, prob_size), dtype=np.float32)
        for b in range(B):
            k = view_lens[last[b]]
            vm[b, rs.choice(k, max(1, int(round(mask_prob * k))), replace=False)] = True
            vprob[b, :k] = soft(k)
        extra.update({'vp_view_mrc_masks': torch.from_numpy(vm), 'vp_view_probs': torch.from_numpy(vprob)})
        if objects > 0:
            omax = int(obj_lens[last].max())
            om = np.zeros((B, omax), dtype=bool)
            oprob = np.zeros((B, omax, prob_size), dtype=np.float32)
            for b in range(B):
                k = obj_lens[last[b]]
                om[b, rs.choice(k, max(1, int(round(mask_prob * k))), replace=False)] = True
                oprob[b, :k] = soft(k)
            extra.update({'vp_obj_mrc_masks': torch.from_numpy(om), 'vp_obj_probs': torch.from_numpy(oprob)})

    t = torch.from_numpy
    out = {
        'txt_ids': t(txt_ids), 'txt_lens': torch.tensor(Ls, dtype=torch.int64), 'txt_labels': t(txt_labels),
        'traj_view_img_fts': t(fts), 'traj_loc_fts': t(loc), 'traj_nav_types': t(nav_types),
        'traj_step_lens': list(Ts), 'traj_vp_view_lens': t(view_lens.astype(np.int64)),
        'traj_vpids': traj_vpids, 'traj_cand_vpids': traj_cand_vpids, 'gmap_vpids': gmap_vpids,
        'gmap_lens': t(gmap_lens), 'gmap_step_ids': t(step_ids), 'gmap_pos_fts': t(pos_fts),
        'gmap_pair_dists': t(pair), 'gmap_visited_masks': t(vis), 'vp_pos_fts': t(vp_pos),
        'global_act_labels': t(global_lab), 'local_act_labels': t(local_lab),
        'extra_heads': [True] * B, 'traj_reverie_loc_fts': None,
    }
    if zdict:                                   # BACL dictionaries (P/data/tasks.py:156-164): (K_direction, K_landmark)
        def pz(k):
            p = rs.uniform(0.1, 1.0, (B, k, 1))
            return torch.from_numpy((p / p.sum(1, keepdims=True)).astype(np.float32))
        kd, kl = zdict
        extra.update({'instr_z_direction_features': torch.from_numpy(rs.uniform(0, 1, (B, kd, 768)).astype(np.float32)),
                      'instr_z_direction_pzs': pz(kd),
                      'instr_z_landmark_features': torch.from_numpy(rs.uniform(0, 1, (B, kl, 768)).astype(np.float32)),
                      'instr_z_landmark_pzs': pz(kl)})
    out.update(extra)
    return out


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


# ======================================================================================= fine-tune episode
def make_nav_episode(B=2, L=44, V=36, n_steps=3, n_cand=4, seed=0, vocab_size=50265, dict_sizes=(35, 39, 50, 24),
                     objects=0, extra_nodes=0):
    """Synthetic stand-in for one DAgger rollout of the fine-tuning loop (M/r2r/agent.py:515-592; shapes of
    M/utils/efficiency_count.py:16-137): text once, then per step a panorama and the graph inputs.  Returns a
    dict of CPU tensors / lists; `run_nav_episode` drives any model exposing `model(mode, batch)`.
    objects > 0: REVERIE-style steps — up to `objects` object tokens after the views of every panorama (at least one
    on the last step), nav type 2, `vp_obj_masks`, and an object-grounding target on the last step.
    extra_nodes: unvisited map nodes seen earlier in the episode that are not candidates of the current step (their image
    embeddings are the zero padding of run_nav_episode) — pads the global map to the G ~ 60 of a long rollout."""
    rs = np.random.RandomState(seed)
    Kd, Kl, Kr, Kf = dict_sizes
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    txt_lens = rs.randint(L // 2, L + 1, B)
    txt_lens[0] = L
    txt_ids = np.ones((B, L), dtype=np.int64)
    for b in range(B):
        txt_ids[b, :txt_lens[b]] = rs.randint(3, vocab_size, txt_lens[b])

    def pz(k):
        p = rs.uniform(0.1, 1.0, (B, k, 1))
        return f32(p / p.sum(1, keepdims=True))
    ep = {
        'txt_ids': torch.from_numpy(txt_ids),
        'txt_masks': torch.from_numpy(np.arange(L)[None, :] < txt_lens[:, None]),
        'instr_z_direction_features': f32(rs.uniform(0, 1, (B, Kd, 768))), 'instr_z_direction_pzs': pz(Kd),
        'instr_z_landmark_features': f32(rs.uniform(0, 1, (B, Kl, 768))), 'instr_z_landmark_pzs': pz(Kl),
        'front_txt_feats': f32(rs.uniform(0, 1, (B, Kf, 768))),
        'front_vp_feats': f32(rs.uniform(0, 1, (B, Kf, 768))), 'front_gmap_feats': f32(rs.uniform(0, 1, (B, Kf, 768))),
        'z_img_features': f32(rs.uniform(0, 1, (B, Kr, 768))), 'z_img_pzs': pz(Kr),
        'steps': [],
    }
    for t in range(n_steps):
        view_lens = rs.randint(V - 4, V + 1, B)
        view_lens[0] = V
        fts = rs.standard_normal((B, V, 768)).astype(np.float32)
        loc = np.stack([_angle_fts(rs, V) for _ in range(B)], 0)
        nav_types = np.zeros((B, V), dtype=np.int64)
        nav_types[:, :n_cand] = 1
        for b in range(B):
            fts[b, view_lens[b]:] = 0
            loc[b, view_lens[b]:] = 0
        G = 2 + (t + 1) + n_cand + extra_nodes        # [stop], [MEM], visited nodes, current candidates (, older unvisited nodes)
        gmap_vpids, vp_cand_vpids = [], []
        gvis = np.zeros((B, G), dtype=bool)
        for b in range(B):
            visited = ['b%d_p%d' % (b, s) for s in range(t + 1)]
            cands = ['b%d_c%d_%d' % (b, t, j) for j in range(n_cand)]
            if t > 0:
                cands[1] = visited[t - 1]             # a back-edge: candidate already visited
            unv = [c for c in cands if c not in visited]
            ids = [None, 'MEM'] + visited + unv + ['b%d_x%d' % (b, k) for k in range(extra_nodes)]
            ids += [None] * (G - len(ids))
            gmap_vpids.append(ids[:G])
            gvis[b, 2:2 + len(visited)] = True
            vp_cand_vpids.append([None, 'MEM'] + cands + [None] * (V - n_cand))
        gmasks = np.ones((B, G), dtype=bool)
        gmasks[:, 1] = False                          # gmap [MEM] slot is masked out (M/r2r/agent.py:209)
        for b in range(B):
            n_valid = sum(1 for x in gmap_vpids[b][2:] if x is not None) + 2
            gmasks[b, n_valid:] = False
        d = rs.uniform(0, 1, (B, G, G)).astype(np.float32)
        d = np.triu(d, 1)
        d = d + d.transpose(0, 2, 1)
        d[:, :2, :] = 0
        d[:, :, :2] = 0
        vp_masks = np.zeros((B, V + 2), dtype=bool)
        vp_nav = np.zeros((B, V + 2), dtype=bool)
        for b in range(B):
            vp_masks[b, :view_lens[b] + 2] = True
            vp_nav[b, 0] = True
            vp_nav[b, 2:2 + n_cand] = True
        target = np.array([2 + (t + 1) + rs.randint(0, max(1, n_cand - 1)) if t + 1 < n_steps else 0 for _ in range(B)])
        obj = {}
        if objects > 0:
            obj_lens = rs.randint(0, objects + 1, B)
            if t + 1 == n_steps:
                obj_lens = np.maximum(obj_lens, 1)
            O = max(1, int(obj_lens.max()))
            tot = view_lens + obj_lens
            W = int(tot.max())
            ofts = rs.standard_normal((B, O, 768)).astype(np.float32)
            onames = rs.randint(0, 45, (B, O)).astype(np.int64)
            loc2 = np.zeros((B, W, 7), dtype=np.float32)
            nav2 = np.zeros((B, W), dtype=np.int64)
            vp_masks = np.zeros((B, W + 2), dtype=bool)
            vp_nav = np.zeros((B, W + 2), dtype=bool)
            vp_obj = np.zeros((B, W + 2), dtype=bool)
            for b in range(B):
                ofts[b, obj_lens[b]:] = 0
                onames[b, obj_lens[b]:] = 0
                loc2[b, :view_lens[b]] = loc[b, :view_lens[b]]
                loc2[b, view_lens[b]:tot[b]] = _angle_fts(rs, int(obj_lens[b]))
                nav2[b, :n_cand] = 1
                nav2[b, view_lens[b]:tot[b]] = 2
                vp_masks[b, :tot[b] + 2] = True
                vp_nav[b, 0] = True
                vp_nav[b, 2:2 + n_cand] = True
                vp_obj[b, 2 + view_lens[b]:2 + tot[b]] = True
                vp_cand_vpids[b] = [None, 'MEM'] + vp_cand_vpids[b][2:2 + n_cand] + [None] * (W - n_cand)
            loc, nav_types = loc2, nav2
            obj = {'reverie_obj_img_fts': f32(ofts), 'reverie_obj_lens': torch.from_numpy(obj_lens.astype(np.int64)),
                   'reverie_obj_names': torch.from_numpy(onames), 'vp_obj_masks': torch.from_numpy(vp_obj),
                   'obj_target': torch.from_numpy(np.array([2 + view_lens[b] + rs.randint(obj_lens[b]) if t + 1 == n_steps else -100
                                                            for b in range(B)], dtype=np.int64))}
            vp_w = W + 2
        else:
            vp_w = V + 2
        ep['steps'].append({
            'view_img_fts': f32(fts), 'loc_fts': f32(loc), 'nav_types': torch.from_numpy(nav_types),
            'view_lens': torch.from_numpy(view_lens.astype(np.int64)),
            'gmap_step_ids': torch.from_numpy(np.tile(np.arange(G), (B, 1)).astype(np.int64) % 5),
            'gmap_pos_fts': f32(rs.standard_normal((B, G, 7))), 'gmap_masks': torch.from_numpy(gmasks),
            'gmap_pair_dists': f32(d), 'gmap_visited_masks': torch.from_numpy(gvis), 'gmap_vpids': gmap_vpids,
            'vp_pos_fts': f32(rs.standard_normal((B, vp_w, 14))), 'vp_masks': torch.from_numpy(vp_masks),
            'vp_nav_masks': torch.from_numpy(vp_nav), 'vp_cand_vpids': vp_cand_vpids,
            'target': torch.from_numpy(target.astype(np.int64)),
            'gmap_cand_view': n_cand, **obj,
        })
    return ep


def run_nav_episode(model, ep, device='cpu', use_bacl=True, use_facl=True, to_float=True, hoist_text_kv=False, hoist_pano=False):
    """language once, then panorama + navigation per step with the [MEM] token carrying `cls_embeds` of the
    previous step (not detached: back-propagation through time, M/r2r/agent.py:592).  Returns (loss, records).
    hoist_pano: the panoramas of ALL steps through the panorama encoder in ONE call (batch T*B) before the first navigation step.
    With teacher forcing the walk — hence every panorama — is known up front and the panorama encoder sees nothing of the
    navigation state (M/r2r/agent.py:548-556 feeds it the observation only), so the embeddings are those of the per-step calls;
    only the [MEM]-carrying navigation steps stay sequential."""
    from collections import defaultdict
    dd = lambda d: defaultdict(lambda: None, d)
    mv = lambda x: x.to(device) if torch.is_tensor(x) else x
    lang = {'txt_ids': mv(ep['txt_ids']), 'txt_masks': mv(ep['txt_masks'])}
    if use_bacl:
        for k in ('instr_z_direction_features', 'instr_z_direction_pzs', 'instr_z_landmark_features', 'instr_z_landmark_pzs'):
            lang[k] = mv(ep[k])
    if use_facl:
        lang['front_txt_feats'] = mv(ep['front_txt_feats'])
    from . import hipops
    B = lang['txt_ids'].shape[0]
    do_hoist = hoist_pano and len({tuple(st['view_img_fts'].shape) for st in ep['steps']}) == 1
    hoisted = None
    def pano_inputs(sts):
        cat = lambda k: mv(sts[0][k]) if len(sts) == 1 else torch.cat([mv(st[k]) for st in sts], 0)
        pin = {'view_img_fts': cat('view_img_fts'), 'loc_fts': cat('loc_fts'), 'nav_types': cat('nav_types'),
               'view_lens': cat('view_lens'), 'already_dropout': True}
        if use_bacl:             # per-sample copies of the dictionary ([B, K, ...]): one per panorama of the joint batch
            tile = lambda z: z if len(sts) == 1 or z.shape[0] != B else z.repeat(len(sts), *([1] * (z.dim() - 1)))
            pin['z_img_features'], pin['z_img_pzs'] = tile(mv(ep['z_img_features'])), tile(mv(ep['z_img_pzs']))
        if 'reverie_obj_img_fts' in sts[0]:
            for k in ('reverie_obj_img_fts', 'reverie_obj_lens', 'reverie_obj_names'):
                pin[k] = cat(k) if torch.is_tensor(sts[0][k]) else sum((list(st[k]) for st in sts), [])
        return pin
    # the hoisted panorama pass sees the observations only: a parallel branch of the instruction encoder when the episode is captured
    with hipops.Branch('pano', 'nav_pano') as bp:
        if do_hoist:
            pa, pm, fu = model('panorama', dd(pano_inputs(ep['steps'])))
    txt = model('language', dd(lang))
    T_ = len(ep['steps'])
    on_gpu = torch.is_tensor(txt) and txt.is_cuda
    # the instruction states (and their hoisted K|V projections) are read by every step: one autograd handle per step, so that the T
    # gradients of each tensor meet in ONE launch (hipops.fanout) instead of T - 1 engine adds of [B, L, 768] / [B, L, 1536] tensors
    txt_h = hipops.fanout(txt, T_ + 1) if on_gpu else [txt] * (T_ + 1)
    txt_kv = model('text_kv', {'txt_embeds': txt_h[T_]}) if hoist_text_kv else None      # (HIP model only: the instruction's K|V once per episode)
    kv_h = hipops.fanout_tree(txt_kv, T_) if (txt_kv is not None and on_gpu) else [txt_kv] * T_
    if do_hoist:
        bp.join(pa, pm, fu)
        # (unbind, not T slices: ONE backward node stacks the per-step gradients instead of T zero-filled slice_backward tensors + T - 1 adds)
        pa_s, pm_s = pa.view(T_, B, *pa.shape[1:]).unbind(0), pm.view(T_, B, *pm.shape[1:]).unbind(0)
        fu_s = [None] * T_ if fu is None else fu.view(T_, B, *fu.shape[1:]).unbind(0)
        hoisted = [(pa_s[i], pm_s[i], fu_s[i]) for i in range(T_)]
    mem = None
    loss = 0.0
    ce_rows = []
    rec = {'txt_embeds': txt, 'steps': []}
    if use_facl and on_gpu:      # the FACL dictionaries are constant over the episode: cast to the compute dtype once, not in every step
        from . import layers
        front_vp, front_gmap = mv(ep['front_vp_feats']).to(layers.compute_dtype()), mv(ep['front_gmap_feats']).to(layers.compute_dtype())
    elif use_facl:
        front_vp, front_gmap = mv(ep['front_vp_feats']), mv(ep['front_gmap_feats'])
    fused_hist = []
    for t, st in enumerate(ep['steps']):
        has_obj = 'reverie_obj_img_fts' in st
        pano, pmask, fused = hoisted[t] if hoisted is not None else model('panorama', dd(pano_inputs([st])))
        fused_hist.append(fused)
        H = pano.shape[-1]
        zero = pano.new_zeros(B, 1, H)
        memtok = zero if mem is None else mem.unsqueeze(1).to(pano.dtype)
        nc = st['gmap_cand_view']
        G = st['gmap_step_ids'].shape[1]
        parts = [zero, memtok] + [f.unsqueeze(1) for f in fused_hist] + [pano[:, :nc]]
        gimg = torch.cat(parts, 1)
        gimg = gimg[:, :G] if gimg.shape[1] >= G else torch.cat([gimg, pano.new_zeros(B, G - gimg.shape[1], H)], 1)
        vimg = torch.cat([zero, memtok, pano], 1)
        nin = {'txt_embeds': txt_h[t], 'txt_masks': mv(ep['txt_masks']), 'gmap_img_embeds': gimg,
               'gmap_step_ids': mv(st['gmap_step_ids']), 'gmap_pos_fts': mv(st['gmap_pos_fts']), 'gmap_masks': mv(st['gmap_masks']),
               'gmap_pair_dists': mv(st['gmap_pair_dists']), 'gmap_visited_masks': mv(st['gmap_visited_masks']),
               'gmap_vpids': st['gmap_vpids'], 'vp_img_embeds': vimg, 'vp_pos_fts': mv(st['vp_pos_fts']),
               'vp_masks': mv(st['vp_masks']), 'vp_nav_masks': mv(st['vp_nav_masks']),
               'vp_obj_masks': mv(st['vp_obj_masks']) if has_obj else None,
               'vp_cand_vpids': st['vp_cand_vpids'], 'flops_count': False, 'nav_fusion': st.get('nav_fusion')}
        if txt_kv is not None:
            nin['txt_kv'] = kv_h[t]
        if use_facl:
            nin['front_vp_feats'], nin['front_gmap_feats'] = front_vp, front_gmap
        out = model('navigation', dd(nin))
        mem = out['cls_embeds']
        logits = out['fused_logits'].float() if to_float else out['fused_logits']
        if logits.is_cuda:      # (one launch per direction; targets of -100 are ignored rows; the T loss vectors are summed once, below)
            ce_rows.append(hipops.cross_entropy_rows(logits, mv(st['target'])))
        else:
            loss = loss + torch.nn.functional.cross_entropy(logits, mv(st['target']), reduction='sum', ignore_index=-100)
        if has_obj and t + 1 == len(ep['steps']):        # object grounding at the stop step (M/reverie/agent_obj.py)
            ol = out['obj_logits'].float() if to_float else out['obj_logits']
            loss = loss + torch.nn.functional.cross_entropy(ol, mv(st['obj_target']), reduction='sum', ignore_index=-100)
        rec['steps'].append({'pano_embeds': pano, 'pano_fused': fused, **out})
    if ce_rows:
        loss = loss + torch.stack(ce_rows, 0).sum()
    return loss, rec


# ------------------------------------------------------------------------------------------------ rollout cases
def rollout_episodes(scan, rs, B=3, max_steps=5, starts=None):
    """B episodes on a rollout.ScanGraph: ground-truth path = shortest path from a start viewpoint to a distant one (cut to
    max_steps viewpoints), random start heading, random instruction ids between <s> (0) and </s> (2)."""
    eps = []
    n = len(scan.vpids)
    starts = starts if starts is not None else [(7 * b * b + 7 * b) % n if b else 0 for b in range(B)]
    starts = [0, 7, 19][:B] if B <= 3 and n > 19 else starts
    dist, _ = scan.shortest()
    for b in range(B):
        s = starts[b] % n
        far = int(np.argsort(dist[s])[-((b % 5) + 2)])
        path = scan.shortest_path(scan.vpids[s], scan.vpids[far])[:max_steps]
        eps.append({'instr_id': 'ep%d' % b, 'scan': scan, 'path': path, 'heading': float(rs.uniform(0, 2 * np.pi)),
                    'instr_encoding': [0] + rs.randint(3, 900, 6 + 3 * (b % 8)).tolist() + [2]})
    return eps


def reverie_episodes(scan, objects, rs, B=3, max_steps=5, starts=None):
    """REVERIE-style episodes (M/reverie/env.py:136-151): rollout_episodes plus a target object — one of the objects the path's last
    viewpoint sees (None where it sees none) — and `end_vps`, the viewpoints the target can be seen from (here: the last viewpoint and,
    for every other episode, its first neighbour as well)."""
    eps = rollout_episodes(scan, rs, B, max_steps, starts)
    for b, ep in enumerate(eps):
        last = ep['path'][-1]
        ids = objects.attrs['%s_%s' % (scan.name, last)]['obj_ids'][:objects.count['%s_%s' % (scan.name, last)]]
        ep['obj_id'] = ids[int(rs.randint(len(ids)))] if len(ids) else None
        ep['end_vps'] = [last] + ([scan.vpids[scan.adj[scan.index[last]][0]]] if b % 2 else [])
    return eps


def make_reverie_rollout_case(seed=23, n_nodes=22, B=3, max_steps=5, scan_seed=12, max_objects=5):
    """make_rollout_case with objects on the viewpoints and a target object per episode (REVERIE): -> scan, features, episodes,
    dictionaries, rollout.ObjectStore (float32 table on the host; `.to(device)` before use)."""
    from . import rollout
    scan, feats, _, dicts = make_rollout_case(seed, n_nodes, B, max_steps, scan_seed)
    objects = rollout.ObjectStore.synthetic([scan], D=768, max_objects=max_objects, seed=seed + 1, dtype=torch.float32, p_empty=0.2)
    eps = reverie_episodes(scan, objects, np.random.RandomState(seed + 2), B, max_steps)
    return scan, feats, eps, dicts, objects


def make_rollout_case(seed=17, n_nodes=24, B=3, max_steps=5, scan_seed=9):
    """scan, float32 features [n_vp, 36, 768], episodes and the BACL / FACL dictionaries (in the reference's on-disk shapes:
    [K, 768] features, [K] probabilities) of the end-to-end rollout golden (tests/golden/rollout_episode.npz).  numpy
    RandomState: bit-stable across machines."""
    from . import rollout
    rs = np.random.RandomState(seed)
    scan = rollout.ScanGraph.synthetic('scanB', n=n_nodes, seed=scan_seed, degree=3)
    feats = rs.standard_normal((len(scan.vpids), 36, 768)).astype(np.float32)
    eps = rollout_episodes(scan, rs, B, max_steps)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

    def pz(k):
        p = rs.uniform(0.1, 1.0, k)
        return f32(p / p.sum())
    dicts = {'instr_direction_features': f32(rs.uniform(0, 1, (35, 768))), 'instr_direction_pzs': pz(35),
             'instr_landmark_features': f32(rs.uniform(0, 1, (39, 768))), 'instr_landmark_pzs': pz(39),
             'img_features': f32(rs.uniform(0, 1, (24, 768))), 'img_pzs': pz(24),
             'txt_feats': rs.uniform(0, 1, (50, 768)).astype(np.float32), 'vp_feats': rs.uniform(0, 1, (50, 768)).astype(np.float32),
             'gmap_feats': rs.uniform(0, 1, (50, 768)).astype(np.float32)}
    return scan, feats, eps, dicts


def rollout_extras(dicts, B, device):
    """the confounder dictionaries as the per-mode extra inputs of rollout.NavRollout.run: what M/r2r/agent.py:53-58,138-140,
    491-511 does with z_dicts / z_front_dict (repeat over the batch)."""
    dev = torch.device(device)
    rep = lambda t, w: t.to(dev).reshape(1, -1, w).repeat(B, 1, 1)
    front = lambda k: torch.from_numpy(dicts[k]).to(dev).unsqueeze(0).repeat(B, 1, 1)
    return {'language': {'instr_z_direction_features': rep(dicts['instr_direction_features'], 768),
                         'instr_z_direction_pzs': rep(dicts['instr_direction_pzs'], 1),
                         'instr_z_landmark_features': rep(dicts['instr_landmark_features'], 768),
                         'instr_z_landmark_pzs': rep(dicts['instr_landmark_pzs'], 1), 'front_txt_feats': front('txt_feats')},
            'panorama': {'z_img_features': rep(dicts['img_features'], 768), 'z_img_pzs': rep(dicts['img_pzs'], 1)},
            'navigation': {'front_txt_feats': front('txt_feats'), 'front_vp_feats': front('vp_feats'), 'front_gmap_feats': front('gmap_feats')}}
