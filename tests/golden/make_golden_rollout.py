"""Golden vectors for the fine-tuning rollout machinery FROM THE IMPORTED REFERENCE (this container only):

  * `GraphMap` / `FloydGraph` of /root/reference/map_nav_src/models/graph_utils.py:43-144 driven over a scripted walk on a synthetic
    scan: distance matrix, `_point`-expanded path lengths, `get_pos_fts`, node-embedding means after the rewrite / accumulate calls of
    the rollout loop (M/r2r/agent.py:549-557);
  * the input builders of `GMapNavAgent` (M/r2r/agent.py): `_panorama_feature_variable_do` (:82-148), `_nav_gmap_variable`
    (:151-237), `_nav_vp_variable_mem` (:271-304), `_teacher_action` (:306-347), called as plain functions on a stand-in `self`
    (args only).  The module imports half a dozen packages this image lacks (MatterSim is not among them — it is imported by
    r2r/env.py only); they are replaced by EMPTY modules for the import, none of their attributes is used by the four builders.
    `.cuda()` is patched to the identity (no GPU in the build container).

The observations come from vln_goat_amd.rollout.GraphSim (the graph-only navigator: MatterSim is absent), converted to the
reference's observation format (feature arrays instead of feature-row numbers).  Output: tests/golden/rollout_walk.npz plus
rollout_walk.json (the id strings).    python tests/golden/make_golden_rollout.py
"""
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

D_FT = 16
B, STEPS = 3, 5


def import_agent():
    ref_shim._install_common()
    for name in ('jsonlines', 'h5py', 'spacy', 'nltk', 'line_profiler', 'sklearnex'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['sklearnex'].patch_sklearn = lambda *a, **k: None
    p = ref_shim.REF_ROOT + '/map_nav_src'
    if p not in sys.path:
        sys.path.insert(0, p)
    torch.Tensor.cuda = lambda self, *a, **k: self
    if not hasattr(np, 'bool'):
        np.bool = bool
    import r2r.agent as agent  # noqa
    import models.graph_utils as gu  # noqa
    return agent, gu


def ref_obs(obs, feats):
    """rollout.GraphSim observation -> the dict M/r2r/env.py:335-377 builds (feature arrays, not rows)."""
    out = []
    for ob in obs:
        ft = feats[ob['feature_row']]                                             # [36, D]
        feature = np.concatenate([ft, ob['view_angle_fts']], -1)
        cands = []
        for c in ob['candidate']:
            cands.append({'viewpointId': c['viewpointId'], 'pointId': c['pointId'], 'heading': c['heading'], 'elevation': c['elevation'],
                          'position': c['position'], 'feature': np.concatenate([ft[c['pointId']], c['angle_feat']], -1)})
        out.append({'instr_id': ob['instr_id'], 'scan': ob['scan'], 'viewpoint': ob['viewpoint'], 'viewIndex': ob['viewIndex'],
                    'position': ob['position'], 'heading': ob['heading'], 'elevation': ob['elevation'], 'feature': feature,
                    'candidate': cands, 'gt_path': ob['gt_path'], 'instr_encoding': ob['instr_encoding']})
    return out


def main():
    agent, gu = import_agent()
    from vln_goat_amd import rollout, synth
    rs = np.random.RandomState(3)
    scan = rollout.ScanGraph.synthetic('scanA', n=28, seed=5, degree=3)
    feats = rs.standard_normal((len(scan.vpids), 36, D_FT)).astype(np.float32)

    class Rows:
        def row(self, s, vp):
            return scan.index[vp]
    sim = rollout.GraphSim(Rows())
    eps = synth.rollout_episodes(scan, rs, B, STEPS)
    obs = sim.reset(eps)
    me = SimpleNamespace(args=SimpleNamespace(image_feat_size=D_FT, act_visited_nodes=False, enc_full_graph=True, ignoreid=-100,
                                              expert_policy='spl'),
                         env=SimpleNamespace(shortest_distances={scan.name: {a: {b: float(scan.shortest()[0][i, j]) for j, b in enumerate(scan.vpids)}
                                                                             for i, a in enumerate(scan.vpids)}}))
    A = agent.GMapNavAgent
    gmaps = [gu.GraphMap(ob['viewpoint']) for ob in obs]
    for g, ob in zip(gmaps, ref_obs(obs, feats)):
        g.update_graph(ob)
    out, ids = {}, {'scan_vpids': scan.vpids, 'paths': [e['path'] for e in eps], 'headings': [e['heading'] for e in eps],
                    'instr': [e['instr_encoding'] for e in eps], 'steps': []}
    out['scan_pos'] = scan.pos
    out['scan_edges'] = np.array([(i, j) for i in range(len(scan.vpids)) for j in scan.adj[i] if i < j], np.int64)
    out['feats'] = feats
    ended = np.zeros(B, bool)
    last = None
    H = 8
    for t in range(STEPS):
        robs = ref_obs(obs, feats)
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.node_step_ids[robs[i]['viewpoint']] = t + 1
        pano = A._panorama_feature_variable_do(me, robs, None, noise=None)
        W = pano['view_img_fts'].shape[1]
        # stand-ins for the panorama encoder's outputs (any differentiable function of the step would do)
        gen = torch.Generator().manual_seed(100 + t)
        pano_embeds = torch.randn(B, W, H, generator=gen)
        fused = torch.randn(B, H, generator=gen)
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.update_node_embed(robs[i]['viewpoint'], fused[i].clone(), rewrite=True)
                for j, cvp in enumerate(pano['cand_vpids'][i]):
                    if not g.graph.visited(cvp):
                        g.update_node_embed(cvp, pano_embeds[i, j].clone())
        nav = A._nav_gmap_variable(me, robs, gmaps, last)
        nav.update(A._nav_vp_variable_mem(me, robs, gmaps, pano_embeds, pano['cand_vpids'], pano['view_lens'], pano['nav_types'], last))
        tgt_il = A._teacher_action(me, robs, nav['gmap_vpids'], ended, visited_masks=nav['gmap_visited_masks'], imitation_learning=True, t=t)
        tgt_spl = A._teacher_action(me, robs, nav['gmap_vpids'], ended, visited_masks=nav['gmap_visited_masks'], imitation_learning=False, t=t,
                                    traj=None)
        k = 't%d_' % t
        for name in ('view_img_fts', 'loc_fts', 'nav_types', 'view_lens'):
            out[k + name] = pano[name].numpy()
        for name in ('gmap_img_embeds', 'gmap_step_ids', 'gmap_pos_fts', 'gmap_visited_masks', 'gmap_pair_dists', 'gmap_masks',
                     'vp_pos_fts', 'vp_masks', 'vp_nav_masks'):
            out[k + name] = nav[name].numpy()
        out[k + 'target_il'], out[k + 'target_spl'] = tgt_il.numpy(), tgt_spl.numpy()
        out[k + 'pano_embeds'], out[k + 'fused'] = pano_embeds.numpy(), fused.numpy()
        # the map itself: distances / path lengths between every pair of nodes in insertion order
        for i, g in enumerate(gmaps):
            names = list(g.node_positions.keys())
            n = len(names)
            Dm, Hm = np.zeros((n, n)), np.zeros((n, n), np.int64)
            for a in range(n):
                for c in range(n):
                    Dm[a, c] = g.graph.distance(names[a], names[c])
                    Hm[a, c] = len(g.graph.path(names[a], names[c]))
            out[k + 'D%d' % i], out[k + 'hops%d' % i] = Dm, Hm
        ids['steps'].append({'cand_vpids': pano['cand_vpids'], 'gmap_vpids': nav['gmap_vpids'], 'vp_cand_vpids': nav['vp_cand_vpids'],
                             'no_vp_left': [bool(x) for x in nav['no_vp_left']],
                             'node_order': [list(g.node_positions.keys()) for g in gmaps],
                             'viewpoints': [ob['viewpoint'] for ob in obs], 'view_index': [int(ob['viewIndex']) for ob in obs]})
        last = torch.randn(B, H, generator=gen)
        out[k + 'last_embeds'] = last.numpy()
        # teacher-forced move (M/r2r/agent.py:592-629 with feedback = 'teacher')
        moves = []
        for i in range(B):
            stop = obs[i]['viewpoint'] == obs[i]['gt_path'][-1]
            if stop or ended[i] or nav['no_vp_left'][i] or t == STEPS - 1:
                moves.append(None)
            else:
                nxt = nav['gmap_vpids'][i][int(tgt_il[i])]
                view = next(c['pointId'] for c in scan.candidates(obs[i]['viewpoint']) if c['viewpointId'] == nxt)
                moves.append((nxt, view))
        obs = sim.step(moves)
        for i, ob in enumerate(ref_obs(obs, feats)):
            if not ended[i]:
                gmaps[i].update_graph(ob)
        ended = np.logical_or(ended, np.array([m is None for m in moves]))
    np.savez_compressed(os.path.join(HERE, 'rollout_walk.npz'), **out)
    with open(os.path.join(HERE, 'rollout_walk.json'), 'w') as f:
        json.dump(ids, f)
    print('wrote rollout_walk.npz (%d arrays), %d steps' % (len(out), len(ids['steps'])))


def main_reverie():
    """REVERIE builders (M/reverie/agent_obj_goat.py): `_panorama_feature_variable_do` (:180-271), `_nav_vp_variable_do` (:345-388),
    `_teacher_action` (:390-417, the shortest-path expert), `_teacher_object` (:419-436) and `ObjectFeatureDB.get_object_feature`
    (M/reverie/data_utils.py:80-99, fed through the class's own feature cache) over a teacher-forced walk with objects on the
    viewpoints.  -> tests/golden/rollout_walk_reverie.npz / .json"""
    import_agent()
    import reverie.agent_obj_goat as ragent
    import reverie.data_utils as rdu
    import models.graph_utils as gu
    from vln_goat_amd import rollout, synth
    rs = np.random.RandomState(4)
    scan = rollout.ScanGraph.synthetic('scanR', n=26, seed=6, degree=3)
    feats = rs.standard_normal((len(scan.vpids), 36, D_FT)).astype(np.float32)
    objects = rollout.ObjectStore.synthetic([scan], D=D_FT, max_objects=6, seed=8, dtype=torch.float32, p_empty=0.25)

    class Rows:
        def row(self, s, vp):
            return scan.index[vp]
    sim = rollout.GraphSim(Rows(), objects=objects)
    eps = synth.reverie_episodes(scan, objects, rs, B, STEPS)
    # the reference's object database with its cache pre-filled (the h5 reader is never reached)
    db = rdu.ObjectFeatureDB.__new__(rdu.ObjectFeatureDB)
    db.obj_feat_size, db._feature_store = D_FT, {}
    otab = objects.table.numpy()
    for key, n in objects.count.items():
        a = objects.attrs[key]
        attrs = {'directions': a['directions'], 'sizes': a['sizes'], 'obj_ids': a['obj_ids'], 'names': a['names']} if n else {}
        db._feature_store[key] = (otab[objects.start[key]:objects.start[key] + n], attrs)

    def ref_obs_reverie(obs):
        out = ref_obs(obs, feats)
        for ro, ob, ep in zip(out, obs, eps):
            f, ang, box, ids, names = db.get_object_feature(ob['scan'], ob['viewpoint'], ob['heading'], ob['elevation'], 4, max_objects=None)
            ro.update({'obj_img_fts': f, 'obj_ang_fts': ang, 'obj_box_fts': box, 'obj_ids': ids, 'obj_name': names,
                       'gt_end_vps': ep['end_vps'], 'gt_obj_id': ep['obj_id']})
        return out
    me = SimpleNamespace(args=SimpleNamespace(image_feat_size=D_FT, act_visited_nodes=False, enc_full_graph=True, ignoreid=-100,
                                              expert_policy='spl'),
                         env=SimpleNamespace(shortest_distances={scan.name: {a: {b: float(scan.shortest()[0][i, j]) for j, b in enumerate(scan.vpids)}
                                                                             for i, a in enumerate(scan.vpids)}}))
    A = ragent.GMapObjectNavAgent
    obs = sim.reset(eps)
    gmaps = [gu.GraphMap(ob['viewpoint']) for ob in obs]
    for g, ob in zip(gmaps, ref_obs_reverie(obs)):
        g.update_graph(ob)
    out = {'scan_pos': scan.pos, 'scan_edges': np.array([(i, j) for i in range(len(scan.vpids)) for j in scan.adj[i] if i < j], np.int64),
           'feats': feats, 'obj_table': otab}
    ids = {'scan_vpids': scan.vpids, 'paths': [e['path'] for e in eps], 'headings': [e['heading'] for e in eps],
           'instr': [e['instr_encoding'] for e in eps], 'obj_id': [e['obj_id'] for e in eps], 'end_vps': [e['end_vps'] for e in eps],
           'objects': {k: {'start': objects.start[k], 'count': objects.count[k], 'directions': objects.attrs[k]['directions'].tolist(),
                           'sizes': objects.attrs[k]['sizes'].tolist(), 'obj_ids': [int(x) for x in objects.attrs[k]['obj_ids']],
                           'names': [int(x) for x in objects.attrs[k]['names']]} for k in objects.count}, 'steps': []}
    ended = np.zeros(B, bool)
    last = None
    H = 8
    for t in range(STEPS):
        robs = ref_obs_reverie(obs)
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.node_step_ids[robs[i]['viewpoint']] = t + 1
        pano = A._panorama_feature_variable_do(me, robs, None, noise=None)
        gen = torch.Generator().manual_seed(300 + t)
        Wp = pano['loc_fts'].shape[1]
        pano_embeds = torch.randn(B, Wp, H, generator=gen)
        fused = torch.randn(B, H, generator=gen)
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.update_node_embed(robs[i]['viewpoint'], fused[i].clone(), rewrite=True)
                for j, cvp in enumerate(pano['cand_vpids'][i]):
                    if not g.graph.visited(cvp):
                        g.update_node_embed(cvp, pano_embeds[i, j].clone())
        nav = A._nav_gmap_variable(me, robs, gmaps, last)
        nav.update(A._nav_vp_variable_do(me, robs, gmaps, pano_embeds, pano['cand_vpids'], pano['view_lens'], pano['reverie_obj_lens'],
                                         pano['nav_types'], last))
        tgt = A._teacher_action(me, robs, nav['gmap_vpids'], ended, visited_masks=nav['gmap_visited_masks'])
        otgt = A._teacher_object(me, robs, ended, pano['view_lens'])
        k = 't%d_' % t
        for name in ('view_img_fts', 'loc_fts', 'nav_types', 'view_lens', 'reverie_obj_img_fts', 'reverie_obj_lens', 'reverie_obj_locs',
                     'reverie_obj_nav_types', 'reverie_obj_names'):
            out[k + name] = pano[name].numpy()
        for name in ('vp_pos_fts', 'vp_masks', 'vp_nav_masks', 'vp_obj_masks', 'gmap_pos_fts', 'gmap_visited_masks', 'gmap_masks'):
            out[k + name] = nav[name].numpy()
        out[k + 'target'], out[k + 'obj_target'] = tgt.numpy(), otgt.numpy()
        ids['steps'].append({'cand_vpids': pano['cand_vpids'], 'obj_ids': [[int(x) for x in o] for o in pano['obj_ids']],
                             'gmap_vpids': nav['gmap_vpids'], 'viewpoints': [ob['viewpoint'] for ob in obs]})
        last = torch.randn(B, H, generator=gen)
        moves = []
        for i in range(B):
            stop = obs[i]['viewpoint'] == obs[i]['gt_path'][-1]
            if stop or ended[i] or nav['no_vp_left'][i] or t == STEPS - 1:
                moves.append(None)
            else:
                nxt = nav['gmap_vpids'][i][int(tgt[i])]
                hop = gmaps[i].graph.path(obs[i]['viewpoint'], nxt)
                prev = obs[i]['viewpoint'] if len(hop) == 1 else hop[-2]
                view = next(c['pointId'] for c in scan.candidates(prev) if c['viewpointId'] == nxt)
                moves.append((nxt, view))
        obs = sim.step(moves)
        for i, ob in enumerate(ref_obs_reverie(obs)):
            if not ended[i]:
                gmaps[i].update_graph(ob)
        ended = np.logical_or(ended, np.array([m is None for m in moves]))
    np.savez_compressed(os.path.join(HERE, 'rollout_walk_reverie.npz'), **out)
    with open(os.path.join(HERE, 'rollout_walk_reverie.json'), 'w') as f:
        json.dump(ids, f)
    print('wrote rollout_walk_reverie.npz (%d arrays), %d steps, object targets %s' % (
        len(out), len(ids['steps']), [out['t%d_obj_target' % t].tolist() for t in range(STEPS)]))


def fingerprint(g):
    if g is None:
        return np.zeros(9, dtype=np.float32)
    flat = g.detach().float().reshape(-1)
    first = torch.zeros(8)
    first[:min(8, flat.numel())] = flat[:8]
    return np.concatenate([[float(flat.double().norm())], first.numpy()]).astype(np.float32)


EP_ARGS = dict(num_l_layers=2, num_x_layers=2, num_pano_layers=2, dropout=0.5, feat_dropout=0.4, do_back_img=True, do_back_txt=True,
               do_front_img=True, do_front_his=True, do_front_txt=True, vocab_size=1200, mode='train', do_back_txt_type='type_2',
               do_back_img_type='type_1', do_add_method='door')
EP_WEIGHT_SEED = 11


def episode_case():
    """The teacher-forced rollout of M/r2r/agent.py:448-676 end to end on the REFERENCE model: reference builders, reference
    GraphMap (embeddings with autograd history: gradients flow through the map into earlier panoramas), BACL + FACL on, loss and
    the gradient fingerprint of every parameter.  -> rollout_episode.npz"""
    agent, gu = import_agent()
    import models.vilmodel_GOAT as vg
    from collections import defaultdict
    from vln_goat_amd import nav_model, rollout, synth
    dd = lambda d: defaultdict(lambda: None, d)
    args = SimpleNamespace(**EP_ARGS)
    cfg = nav_model.nav_config_from_args(args)
    torch.manual_seed(0)
    ref = vg.GlocalTextPathNavCMT(cfg)
    ours = nav_model.GlocalTextPathNavCMT(cfg)
    ref.load_state_dict(synth.seeded_state_dict(ours, seed=EP_WEIGHT_SEED))
    ref.eval()
    scan, feats, eps, dicts = synth.make_rollout_case()

    class Rows:
        def row(self, s, vp):
            return scan.index[vp]
    sim = rollout.GraphSim(Rows())
    obs = sim.reset(eps)
    Bn = len(obs)
    me = SimpleNamespace(args=SimpleNamespace(image_feat_size=768, act_visited_nodes=False, enc_full_graph=True, ignoreid=-100, expert_policy='spl'))
    A = agent.GMapNavAgent
    robs = ref_obs(obs, feats)
    gmaps = [gu.GraphMap(ob['viewpoint']) for ob in robs]
    for g, ob in zip(gmaps, robs):
        g.update_graph(ob)
    instr_zdict = {k: dicts[k] for k in ('instr_direction_features', 'instr_direction_pzs', 'instr_landmark_features', 'instr_landmark_pzs')}
    img_zdict = {'img_features': dicts['img_features'], 'img_pzs': dicts['img_pzs']}
    front = {k: torch.from_numpy(np.array([dicts[k]] * Bn)) for k in ('txt_feats', 'vp_feats', 'gmap_feats')}
    lang = A._language_variable(me, robs, instr_zdict, front['txt_feats'])
    txt_embeds = ref('language', dd(lang))
    ended = np.zeros(Bn, bool)
    last = None
    ml_loss = 0.0
    store = {}
    max_len = 6
    for t in range(max_len):
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.node_step_ids[robs[i]['viewpoint']] = t + 1
        pano = A._panorama_feature_variable_do(me, robs, img_zdict, noise=None)
        pano_embeds, pano_masks, fused = ref('panorama', dd(pano))
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.update_node_embed(robs[i]['viewpoint'], fused[i], rewrite=True)
                for j, cvp in enumerate(pano['cand_vpids'][i]):
                    if not g.graph.visited(cvp):
                        g.update_node_embed(cvp, pano_embeds[i, j])
        nav = A._nav_gmap_variable(me, robs, gmaps, last)
        nav.update(A._nav_vp_variable_mem(me, robs, gmaps, pano_embeds, pano['cand_vpids'], pano['view_lens'], pano['nav_types'], last))
        nav.update({'txt_embeds': txt_embeds, 'txt_masks': lang['txt_masks'], 'front_txt_feats': front['txt_feats'],
                    'front_vp_feats': front['vp_feats'], 'front_gmap_feats': front['gmap_feats']})
        out = ref('navigation', dd(nav))
        last = out['cls_embeds']
        logits = out['fused_logits']
        tgt = A._teacher_action(me, robs, nav['gmap_vpids'], ended, visited_masks=nav['gmap_visited_masks'], imitation_learning=True, t=t)
        ml_loss = ml_loss + torch.nn.functional.cross_entropy(logits, tgt, reduction='sum', ignore_index=-100)
        store['s%d_fused_logits' % t] = logits.detach().numpy()
        store['s%d_cls_embeds' % t] = out['cls_embeds'].detach().numpy()
        store['s%d_target' % t] = tgt.numpy()
        store['s%d_gmap_img_embeds' % t] = nav['gmap_img_embeds'][:, :, :16].detach().numpy()
        moves = []
        for i in range(Bn):
            stop = obs[i]['viewpoint'] == obs[i]['gt_path'][-1]
            if stop or ended[i] or nav['no_vp_left'][i] or t == max_len - 1:
                moves.append(None)
            else:
                nxt = nav['gmap_vpids'][i][int(tgt[i])]
                view = next(c['pointId'] for c in scan.candidates(obs[i]['viewpoint']) if c['viewpointId'] == nxt)
                moves.append((nxt, view))
        obs = sim.step(moves)
        robs = ref_obs(obs, feats)
        for i, ob in enumerate(robs):
            if not ended[i]:
                gmaps[i].update_graph(ob)
        ended = np.logical_or(ended, np.array([m is None for m in moves]))
        if ended.all():
            break
    loss = ml_loss * 1.0 / Bn
    loss.backward()
    store['n_steps'] = np.array([t + 1])
    store['loss'] = np.array([float(loss)], np.float32)
    store['param_names'] = np.array([n for n, _ in ref.named_parameters()])
    store['grad_fp'] = np.stack([fingerprint(p.grad) for _, p in ref.named_parameters()])
    store['txt_embeds'] = txt_embeds[:, :, :16].detach().numpy()
    path = os.path.join(HERE, 'rollout_episode.npz')
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB  loss', float(loss), 'steps', t + 1)


def reverie_episode_case():
    """The REVERIE rollout (M/reverie/agent_obj_goat.py:560-790) end to end on the REFERENCE model (dataset = 'reverie': object tokens,
    object-grounding head) with the REVERIE agent's builders: teacher feedback with the shortest-path expert, navigation loss + object
    grounding loss, per-step navigation and object logits, the gradient fingerprint and random projections of every parameter.
    -> rollout_episode_reverie.npz"""
    agent, gu = import_agent()
    import reverie.agent_obj_goat as ragent
    import reverie.data_utils as rdu
    import models.vilmodel_GOAT as vg
    from collections import defaultdict
    from vln_goat_amd import nav_model, rollout, synth
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from helpers import projections
    dd = lambda d: defaultdict(lambda: None, d)
    args = SimpleNamespace(**{**EP_ARGS, 'dataset': 'reverie', 'obj_feat_size': 768})
    cfg = nav_model.nav_config_from_args(args)
    torch.manual_seed(0)
    ref = vg.GlocalTextPathNavCMT(cfg)
    ours = nav_model.GlocalTextPathNavCMT(cfg)
    ref.load_state_dict(synth.seeded_state_dict(ours, seed=EP_WEIGHT_SEED))
    ref.eval()
    scan, feats, eps, dicts, objects = synth.make_reverie_rollout_case()

    class Rows:
        def row(self, s, vp):
            return scan.index[vp]
    sim = rollout.GraphSim(Rows(), objects=objects)
    db = rdu.ObjectFeatureDB.__new__(rdu.ObjectFeatureDB)
    db.obj_feat_size, db._feature_store = 768, {}
    otab = objects.table.numpy()
    for key, n in objects.count.items():
        a = objects.attrs[key]
        db._feature_store[key] = (otab[objects.start[key]:objects.start[key] + n],
                                  {'directions': a['directions'], 'sizes': a['sizes'], 'obj_ids': a['obj_ids'], 'names': a['names']} if n else {})

    def robs_of(obs):
        out = ref_obs(obs, feats)
        for ro, ob, ep in zip(out, obs, eps):
            f, ang, box, ids, names = db.get_object_feature(ob['scan'], ob['viewpoint'], ob['heading'], ob['elevation'], 4, max_objects=None)
            ro.update({'obj_img_fts': f, 'obj_ang_fts': ang, 'obj_box_fts': box, 'obj_ids': ids, 'obj_name': names,
                       'gt_end_vps': ep['end_vps'], 'gt_obj_id': ep['obj_id']})
        return out
    obs = sim.reset(eps)
    Bn = len(obs)
    me = SimpleNamespace(args=SimpleNamespace(image_feat_size=768, act_visited_nodes=False, enc_full_graph=True, ignoreid=-100, expert_policy='spl'),
                         env=SimpleNamespace(shortest_distances={scan.name: {a: {b: float(scan.shortest()[0][i, j]) for j, b in enumerate(scan.vpids)}
                                                                             for i, a in enumerate(scan.vpids)}}))
    A, R2R = ragent.GMapObjectNavAgent, agent.GMapNavAgent
    robs = robs_of(obs)
    gmaps = [gu.GraphMap(ob['viewpoint']) for ob in robs]
    for g, ob in zip(gmaps, robs):
        g.update_graph(ob)
    instr_zdict = {k: dicts[k] for k in ('instr_direction_features', 'instr_direction_pzs', 'instr_landmark_features', 'instr_landmark_pzs')}
    img_zdict = {'img_features': dicts['img_features'], 'img_pzs': dicts['img_pzs']}
    front = {k: torch.from_numpy(np.array([dicts[k]] * Bn)) for k in ('txt_feats', 'vp_feats', 'gmap_feats')}
    # (both instruction dictionaries, through the R2R agent's language builder: the model's type_2 intervention needs the direction
    #  dictionary, which the REVERIE agent's own `_language_variable` does not pass)
    lang = R2R._language_variable(me, robs, instr_zdict, front['txt_feats'])
    txt_embeds = ref('language', dd(lang))
    ended = np.zeros(Bn, bool)
    last = None
    ml_loss = og_loss = 0.0
    store = {}
    max_len = 6
    for t in range(max_len):
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.node_step_ids[robs[i]['viewpoint']] = t + 1
        pano = A._panorama_feature_variable_do(me, robs, img_zdict, noise=None)
        pano_embeds, pano_masks, fused = ref('panorama', dd(pano))
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.update_node_embed(robs[i]['viewpoint'], fused[i], rewrite=True)
                for j, cvp in enumerate(pano['cand_vpids'][i]):
                    if not g.graph.visited(cvp):
                        g.update_node_embed(cvp, pano_embeds[i, j])
        nav = A._nav_gmap_variable(me, robs, gmaps, last)
        nav.update(A._nav_vp_variable_do(me, robs, gmaps, pano_embeds, pano['cand_vpids'], pano['view_lens'], pano['reverie_obj_lens'],
                                         pano['nav_types'], last))
        nav.update({'txt_embeds': txt_embeds, 'txt_masks': lang['txt_masks'], 'front_txt_feats': front['txt_feats'],
                    'front_vp_feats': front['vp_feats'], 'front_gmap_feats': front['gmap_feats']})
        out = ref('navigation', dd(nav))
        last = out['cls_embeds']
        logits = out['fused_logits']
        tgt = A._teacher_action(me, robs, nav['gmap_vpids'], ended, visited_masks=nav['gmap_visited_masks'])
        otgt = A._teacher_object(me, robs, ended, pano['view_lens'])
        ml_loss = ml_loss + torch.nn.functional.cross_entropy(logits, tgt, reduction='sum', ignore_index=-100)
        og_loss = og_loss + torch.nn.functional.cross_entropy(out['obj_logits'], otgt, reduction='sum', ignore_index=-100)
        store['s%d_fused_logits' % t] = logits.detach().numpy()
        store['s%d_obj_logits' % t] = out['obj_logits'].detach().numpy()
        store['s%d_cls_embeds' % t] = out['cls_embeds'].detach().numpy()
        store['s%d_target' % t], store['s%d_obj_target' % t] = tgt.numpy(), otgt.numpy()
        moves = []
        for i in range(Bn):
            stop = obs[i]['viewpoint'] == obs[i]['gt_path'][-1]
            if stop or ended[i] or nav['no_vp_left'][i] or t == max_len - 1:
                moves.append(None)
            else:
                nxt = nav['gmap_vpids'][i][int(tgt[i])]
                hop = gmaps[i].graph.path(obs[i]['viewpoint'], nxt)
                prev = obs[i]['viewpoint'] if len(hop) == 1 else hop[-2]
                moves.append((nxt, next(c['pointId'] for c in scan.candidates(prev) if c['viewpointId'] == nxt)))
        obs = sim.step(moves)
        robs = robs_of(obs)
        for i, ob in enumerate(robs):
            if not ended[i]:
                gmaps[i].update_graph(ob)
        ended = np.logical_or(ended, np.array([m is None for m in moves]))
        if ended.all():
            break
    loss = ml_loss * 1.0 / Bn + og_loss * 1.0 / Bn
    loss.backward()
    store['n_steps'] = np.array([t + 1])
    store['loss'] = np.array([float(loss)], np.float32)
    store['ml_loss'] = np.array([float(ml_loss) / Bn], np.float32)
    store['og_loss'] = np.array([float(og_loss) / Bn], np.float32)
    store['param_names'] = np.array([n for n, _ in ref.named_parameters()])
    store['grad_fp'] = np.stack([fingerprint(p.grad) for _, p in ref.named_parameters()])
    store['grad_proj'] = np.stack([projections(p.grad) for _, p in ref.named_parameters()])
    path = os.path.join(HERE, 'rollout_episode_reverie.npz')
    np.savez_compressed(path, **store)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB  loss', float(loss), 'ml', float(ml_loss) / Bn, 'og', float(og_loss) / Bn, 'steps', t + 1,
          'object targets', [store['s%d_obj_target' % k].tolist() for k in range(t + 1)])


def scripted_actions(t, nav, ended, rs):
    """The FIXED 'sample' sequence of the sampled-rollout golden: per sample a valid map slot drawn by a seeded numpy generator —
    [stop] (slot 0) with probability 0.12 from step 2 on, else uniform over the unvisited, unmasked nodes (what a sampled policy can
    pick: M/r2r/agent.py masks visited nodes out of the logits).  The test replays this sequence through NavRollout's `sampler` hook."""
    masks = (nav['gmap_masks'] & ~nav['gmap_visited_masks']).numpy()
    acts = np.zeros(len(ended), np.int64)
    for i in range(len(ended)):
        cand = [j for j in np.nonzero(masks[i])[0] if j >= 2]
        u = rs.uniform()
        acts[i] = 0 if (not cand or (t >= 2 and u < 0.12)) else cand[int(rs.randint(len(cand)))]
    return acts


def sample_episode_case():
    """The SAMPLED rollout of the DAgger iteration (feedback = 'sample', M/r2r/agent.py:436-437,629-631,650-672) on the REFERENCE model with a
    FIXED action sequence in place of Categorical.sample(): the policy walks off the ground-truth path, the loss is the cross-entropy
    against `_teacher_action(imitation_learning=False)` (the shortest-path expert of the CURRENT state) at every visited state, a sampled
    [stop] ends an episode without the stop-node backtrack, the forced ends (goal reached / no viewpoint left / last step) go back to the
    node with the best recorded stop probability.  -> rollout_episode_sample.npz (+ .json: the id strings)"""
    agent, gu = import_agent()
    import models.vilmodel_GOAT as vg
    from collections import defaultdict
    from vln_goat_amd import nav_model, rollout, synth
    dd = lambda d: defaultdict(lambda: None, d)
    args = SimpleNamespace(**EP_ARGS)
    cfg = nav_model.nav_config_from_args(args)
    torch.manual_seed(0)
    ref = vg.GlocalTextPathNavCMT(cfg)
    ours = nav_model.GlocalTextPathNavCMT(cfg)
    ref.load_state_dict(synth.seeded_state_dict(ours, seed=EP_WEIGHT_SEED))
    ref.eval()
    scan, feats, eps, dicts = synth.make_rollout_case()

    class Rows:
        def row(self, s, vp):
            return scan.index[vp]
    sim = rollout.GraphSim(Rows())
    obs = sim.reset(eps)
    Bn = len(obs)
    dist = scan.shortest()[0]
    me = SimpleNamespace(args=SimpleNamespace(image_feat_size=768, act_visited_nodes=False, enc_full_graph=True, ignoreid=-100, expert_policy='spl'),
                         env=SimpleNamespace(shortest_distances={scan.name: {a: {b: float(dist[i, j]) for j, b in enumerate(scan.vpids)}
                                                                             for i, a in enumerate(scan.vpids)}}))
    A = agent.GMapNavAgent
    robs = ref_obs(obs, feats)
    gmaps = [gu.GraphMap(ob['viewpoint']) for ob in robs]
    for g, ob in zip(gmaps, robs):
        g.update_graph(ob)
    instr_zdict = {k: dicts[k] for k in ('instr_direction_features', 'instr_direction_pzs', 'instr_landmark_features', 'instr_landmark_pzs')}
    img_zdict = {'img_features': dicts['img_features'], 'img_pzs': dicts['img_pzs']}
    front = {k: torch.from_numpy(np.array([dicts[k]] * Bn)) for k in ('txt_feats', 'vp_feats', 'gmap_feats')}
    lang = A._language_variable(me, robs, instr_zdict, front['txt_feats'])
    txt_embeds = ref('language', dd(lang))
    ended = np.zeros(Bn, bool)
    just_ended = np.zeros(Bn, bool)
    traj = [{'instr_id': ob['instr_id'], 'path': [[ob['viewpoint']]]} for ob in robs]
    last = None
    ml_loss = 0.0
    store, ids = {}, {'actions': [], 'viewpoints': []}
    max_len = 7
    rs = np.random.RandomState(23)
    for t in range(max_len):
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.node_step_ids[robs[i]['viewpoint']] = t + 1
        pano = A._panorama_feature_variable_do(me, robs, img_zdict, noise=None)
        pano_embeds, pano_masks, fused = ref('panorama', dd(pano))
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.update_node_embed(robs[i]['viewpoint'], fused[i], rewrite=True)
                for j, cvp in enumerate(pano['cand_vpids'][i]):
                    if not g.graph.visited(cvp):
                        g.update_node_embed(cvp, pano_embeds[i, j])
        nav = A._nav_gmap_variable(me, robs, gmaps, last)
        nav.update(A._nav_vp_variable_mem(me, robs, gmaps, pano_embeds, pano['cand_vpids'], pano['view_lens'], pano['nav_types'], last))
        nav.update({'txt_embeds': txt_embeds, 'txt_masks': lang['txt_masks'], 'front_txt_feats': front['txt_feats'],
                    'front_vp_feats': front['vp_feats'], 'front_gmap_feats': front['gmap_feats']})
        out = ref('navigation', dd(nav))
        last = out['cls_embeds']
        logits = out['fused_logits']
        probs = torch.softmax(logits, 1)
        for i, g in enumerate(gmaps):                        # M/r2r/agent.py:605-611
            if not ended[i]:
                g.node_stop_scores[robs[i]['viewpoint']] = {'stop': probs[i, 0].data.item()}
        tgt = A._teacher_action(me, robs, nav['gmap_vpids'], ended, visited_masks=nav['gmap_visited_masks'], imitation_learning=False, t=t, traj=traj)
        ml_loss = ml_loss + torch.nn.functional.cross_entropy(logits, tgt, reduction='sum', ignore_index=-100)
        a_t = scripted_actions(t, nav, ended, rs)
        store['s%d_fused_logits' % t] = logits.detach().numpy()
        store['s%d_cls_embeds' % t] = out['cls_embeds'].detach().numpy()
        store['s%d_target' % t] = tgt.numpy()
        store['s%d_action' % t] = a_t
        ids['viewpoints'].append([ob['viewpoint'] for ob in robs])
        a_t_stop = [ob['viewpoint'] == ob['gt_path'][-1] for ob in robs]           # :650-651 (training feedback)
        cpu_a_t = []
        for i in range(Bn):                                                          # :656-662
            if a_t_stop[i] or ended[i] or nav['no_vp_left'][i] or (t == max_len - 1):
                cpu_a_t.append(None)
                just_ended[i] = True
            else:
                cpu_a_t.append(nav['gmap_vpids'][i][int(a_t[i])])
        moves = []
        for i, vp in enumerate(cpu_a_t):                                             # make_equiv_action (:349-384) on the graph-only navigator
            if vp is None:
                moves.append(None)
                continue
            hop = gmaps[i].graph.path(robs[i]['viewpoint'], vp)
            traj[i]['path'].append(hop)
            prev = traj[i]['path'][-2][-1] if len(hop) == 1 else hop[-2]
            view = next(c['pointId'] for c in scan.candidates(prev) if c['viewpointId'] == vp)
            moves.append((vp, view))
        for i in range(Bn):                                                          # :665-672
            if (not ended[i]) and just_ended[i]:
                stop_node, stop_score = None, {'stop': -float('inf')}
                for k, v in gmaps[i].node_stop_scores.items():
                    if v['stop'] > stop_score['stop']:
                        stop_score, stop_node = v, k
                if stop_node is not None and robs[i]['viewpoint'] != stop_node:
                    traj[i]['path'].append(gmaps[i].graph.path(robs[i]['viewpoint'], stop_node))
        obs = sim.step(moves)
        robs = ref_obs(obs, feats)
        for i, ob in enumerate(robs):
            if not ended[i]:
                gmaps[i].update_graph(ob)
        ended = np.logical_or(ended, np.array([x is None for x in cpu_a_t]))
        if ended.all():
            break
    loss = ml_loss * 1.0 / Bn
    loss.backward()
    store['n_steps'] = np.array([t + 1])
    store['loss'] = np.array([float(loss)], np.float32)
    store['param_names'] = np.array([n for n, _ in ref.named_parameters()])
    store['grad_fp'] = np.stack([fingerprint(p.grad) for _, p in ref.named_parameters()])
    ids['traj'] = [tr['path'] for tr in traj]
    path = os.path.join(HERE, 'rollout_episode_sample.npz')
    np.savez_compressed(path, **store)
    with open(os.path.join(HERE, 'rollout_episode_sample.json'), 'w') as f:
        json.dump(ids, f)
    off_path = sum(1 for tr, ep in zip(traj, eps) if [h[-1] for h in tr['path']][:len(ep['path'])] != ep['path'])
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB  loss', float(loss), 'steps', t + 1, ' episodes off the ground-truth path:', off_path,
          ' actions', [store['s%d_action' % k].tolist() for k in range(t + 1)])


if __name__ == '__main__':
    if sys.argv[1:] == ['sample']:
        sample_episode_case()
    elif sys.argv[1:] == ['episode']:
        episode_case()
    elif sys.argv[1:] == ['reverie']:
        main_reverie()
        reverie_episode_case()
    else:
        main()
        if not sys.argv[1:]:
            episode_case()
            sample_episode_case()
            main_reverie()
            reverie_episode_case()
