"""Fine-tuning rollout without MatterSim (SURVEY §8f N4 + the fine-tuning half of N1).

The reference drives its fine-tuning loop through MatterSim with rendering switched OFF (M/r2r/env.py:47-58): the simulator is
used as a graph walker only — adjacency, headings, the 36 discretised view directions — and everything the model sees is
assembled on the host by Python loops over viewpoint-id strings (M/r2r/agent.py:82-304) around a per-episode `GraphMap`
(M/models/graph_utils.py:43-144) whose incremental Floyd update is an O(N^2) Python double loop per step.

Here (M/ = /root/reference/map_nav_src):

  * `ScanGraph` / `GraphSim`      the graph-only navigator: connectivity JSON (or a synthetic scan) -> positions, adjacency,
                                  per-viewpoint candidate tables (M/r2r/env.py:241-333 `make_candidate` without the simulator),
                                  observations of `_get_obs` (:335-377), teleport steps of `make_equiv_action` (M/r2r/agent.py:349-378).
  * `FloydGraph` / `GraphMap`     the reference's map with the SAME update rule (single-pivot relaxation per visited node, the
                                  95959595 default distance, lazily evaluated `_point` paths) as vectorised float64 numpy — decisions
                                  are bit-identical to the reference's Python floats (tests/golden/rollout_walk.npz).
  * `NodeEmbedStore`              the node-embedding half of GraphMap on the DEVICE: panorama outputs of every step stay in HBM as a
                                  pool; "rewrite" / "sum + count" / "mean on read" (graph_utils.py:113-125) become a CSR gather /
                                  segment-mean over that pool (goat_gather_segmean_*), differentiable, so the gradient of a later
                                  step's map tokens reaches the panorama encoder of the step that produced them — as it does in the
                                  reference through `pad_tensors_wgrad` (M/r2r/agent.py:211).
  * `panorama_inputs`, `gmap_inputs`, `vp_inputs`, `teacher_action`   the builders of M/r2r/agent.py:82-148,151-237,271-304,306-347,
                                  producing the `panorama` / `navigation` input dicts of VLNBert.forward — small integer / float32
                                  tables built by numpy on the host; the 36 x 768 view features are never touched on the host: the
                                  builders emit ROW INDICES into a device-resident feature table (features.FeatureStore) and one
                                  gather kernel assembles the batch in HBM.
  * `NavRollout`                  the rollout loop (M/r2r/agent.py:448-676) for feedback = teacher / argmax / sample.  Teacher
                                  forcing needs no device->host copy at all (the next action comes from the ground-truth path), so the
                                  host builds step t+1 while the GPU runs step t; sampled rollouts read back B action indices per
                                  step, as the reference does.

MatterSim itself is absent from this image: the candidate ORDER inside a panorama (first view index in which a neighbour falls
inside the camera frustum, then angular distance) restates the simulator's documented behaviour and is pinned by this repo's own
fixtures only; everything downstream of the observations is pinned to outputs of the imported reference
(tests/golden/make_golden_rollout.py)."""
import json
import math
import os

import numpy as np
import torch

MAX_DIST = 30          # M/models/graph_utils.py:4-5
MAX_STEP = 10
FLOYD_INF = 95959595   # graph_utils.py:45 (the reference's "not connected" distance; also the never-written diagonal)
HFOV = math.radians(80.0)      # 640 x 480 at VFOV 60 (M/r2r/env.py:41-44)
VFOV = math.radians(60.0)


# ------------------------------------------------------------------------------------------------ geometry (M/utils/data.py)
def angle_feature(heading, elevation, angle_feat_size=4):
    # M/utils/data.py:128-131
    return np.array([math.sin(heading), math.cos(heading), math.sin(elevation), math.cos(elevation)] * (angle_feat_size // 4), dtype=np.float32)


def get_angle_fts(headings, elevations, angle_feat_size=4):
    # M/utils/data.py:177-183 (sin / cos of the float32 angles)
    headings, elevations = np.asarray(headings), np.asarray(elevations)
    ang = np.empty((headings.shape[0], 4), np.float32)
    ang[:, 0], ang[:, 1], ang[:, 2], ang[:, 3] = np.sin(headings), np.cos(headings), np.sin(elevations), np.cos(elevations)
    reps = angle_feat_size // 4
    return np.concatenate([ang] * reps, 1) if reps > 1 else ang


def view_angles(view_index):
    """heading, elevation of discretised view 0..35 (12 headings x 3 elevations, M/r2r/env.py:71-74)."""
    return (view_index % 12) * math.radians(30), (view_index // 12 - 1) * math.radians(30)


def view_angle_feature_table(angle_feat_size=4):
    """[36 base views][36 views, angle_feat_size]: get_all_point_angle_feature (M/utils/data.py:133-156) — the simulator is only
    used there to enumerate the 36 (heading, elevation) pairs."""
    out = np.empty((36, 36, angle_feat_size), np.float32)
    for base in range(36):
        bh, be = view_angles(base)
        for ix in range(36):
            h, e = view_angles(ix)
            out[base, ix] = angle_feature(h - bh, e - be, angle_feat_size)
    return out


def rel_pos(a, b):
    """absolute heading / elevation / distance of points b [n,3] seen from a [3] (calculate_vp_rel_pos_fts, M/utils/data.py:158-175,
    before the base angles are subtracted), float64."""
    b = np.asarray(b, np.float64).reshape(-1, 3)
    a = np.asarray(a, np.float64)
    dx, dy, dz = b[:, 0] - a[0], b[:, 1] - a[1], b[:, 2] - a[2]
    xy = np.maximum(np.sqrt(dx ** 2 + dy ** 2), 1e-8)
    xyz = np.maximum(np.sqrt(dx ** 2 + dy ** 2 + dz ** 2), 1e-8)
    heading = np.arcsin(dx / xy)
    heading = np.where(b[:, 1] < a[1], np.pi - heading, heading)
    elevation = np.arcsin(dz / xyz)
    return heading, elevation, xyz


# ------------------------------------------------------------------------------------------------ the map (graph_utils.py)
class FloydGraph:
    """M/models/graph_utils.py:43-88 on dense float64 matrices.  `update(k)` is the reference's single-pivot relaxation: inside
    its double loop only entries [x][k] and [k][y] are read and neither can improve (the diagonal keeps the 95959595 default), so
    the loop is order-independent and one vectorised min; `_point` is kept as an index matrix and paths are expanded lazily from
    its CURRENT state, exactly as `path()` recurses in the reference."""

    def __init__(self, cap=32):
        self.ids, self.names = {}, []
        self.D = np.full((cap, cap), float(FLOYD_INF))
        self.P = np.full((cap, cap), -1, np.int32)
        self._visited = set()
        self._hop_memo = {}         # (i, j) -> hops under the CURRENT _point matrix (cleared whenever an entry of P changes)

    def _ix(self, vp):
        i = self.ids.get(vp)
        if i is None:
            i = self.ids[vp] = len(self.names)
            self.names.append(vp)
            if i >= self.D.shape[0]:
                cap = 2 * self.D.shape[0]
                D = np.full((cap, cap), float(FLOYD_INF))
                P = np.full((cap, cap), -1, np.int32)
                D[:i, :i], P[:i, :i] = self.D[:i, :i], self.P[:i, :i]
                self.D, self.P = D, P
        return i

    def distance(self, x, y):
        if x == y:
            return 0
        return self.D[self._ix(x), self._ix(y)]

    def add_edge(self, x, y, dis):
        i, j = self._ix(x), self._ix(y)
        if dis < self.D[i, j]:
            self.D[i, j] = self.D[j, i] = dis
            self.P[i, j] = self.P[j, i] = -1
            self._hop_memo.clear()

    def update(self, k):
        kk, n = self._ix(k), len(self.names)
        D, P = self.D[:n, :n], self.P[:n, :n]
        cand = D[:, kk][:, None] + D[kk, :][None, :]
        better = cand < D
        np.fill_diagonal(better, False)
        D[better] = cand[better]
        P[better] = kk
        self._hop_memo.clear()
        self._visited.add(k)

    def visited(self, k):
        return k in self._visited

    def _hops(self, i, j, depth=0):
        if i == j:
            return 0
        memo = self._hop_memo
        n = memo.get((i, j))
        if n is None:               # (sub-paths are shared between the pairs a step asks for)
            k = int(self.P[i, j])
            if k < 0:
                n = 1
            else:
                if depth > 4096:
                    raise RecursionError('FloydGraph: cyclic _point chain')
                n = self._hops(i, k, depth + 1) + self._hops(k, j, depth + 1)
            memo[(i, j)] = n
        return n

    def path_len(self, x, y):
        return self._hops(self._ix(x), self._ix(y))

    def path(self, x, y):
        if x == y:
            return []
        i, j = self._ix(x), self._ix(y)
        k = self.P[i, j]
        if k < 0:
            return [y]
        return self.path(x, self.names[k]) + self.path(self.names[k], y)

    def dist_rows(self, x, ys):
        """distance(x, y) for every y of ys (vectorised read; 0 where y == x)."""
        i = self._ix(x)
        j = np.array([self._ix(y) for y in ys], dtype=np.int64)
        d = self.D[i, j].copy()
        d[j == i] = 0
        return d


class GraphMap:
    """M/models/graph_utils.py:91-144 without the embeddings (those live on the device: NodeEmbedStore)."""

    def __init__(self, start_vp):
        self.start_vp = start_vp
        self.node_positions = {}
        self.graph = FloydGraph()
        self.node_stop_scores = {}
        self.node_step_ids = {}

    def update_graph(self, ob):
        self.node_positions[ob['viewpoint']] = ob['position']
        p = np.asarray(ob['position'], np.float64)
        for cc in ob['candidate']:
            self.node_positions[cc['viewpointId']] = cc['position']
            q = np.asarray(cc['position'], np.float64)
            d = q - p
            dist = np.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2)          # calc_position_distance, :7-13
            self.graph.add_edge(ob['viewpoint'], cc['viewpointId'], dist)
        self.graph.update(ob['viewpoint'])

    def get_pos_fts(self, cur_vp, gmap_vpids, cur_heading, cur_elevation, angle_feat_size=4):
        """[n, angle_feat_size + 3]: sin / cos of the relative heading and elevation, line distance, map distance, map path length
        (graph_utils.py:127-149); None entries ([stop] / [MEM]) give the angle features of (0, 0) and zero distances."""
        n = len(gmap_vpids)
        first = 0
        while first < n and gmap_vpids[first] is None:
            first += 1
        vps = gmap_vpids[first:]
        if None in vps:             # (None entries between real nodes: not what the builders make, kept general)
            real = [i for i, vp in enumerate(gmap_vpids) if vp is not None]
            vps = [gmap_vpids[i] for i in real]
        else:
            real = slice(first, n)
        ang = np.zeros((n, 2), np.float64)
        dist = np.zeros((n, 3), np.float64)
        if vps:
            graph = self.graph
            pos = np.array([self.node_positions[vp] for vp in vps], np.float64)
            h, e, d = rel_pos(self.node_positions[cur_vp], pos)
            ang[real, 0], ang[real, 1] = h - cur_heading, e - cur_elevation
            dist[real, 0] = d / MAX_DIST
            dist[real, 1] = graph.dist_rows(cur_vp, vps) / MAX_DIST
            dist[real, 2] = np.array([graph.path_len(cur_vp, vp) for vp in vps], np.float64) / MAX_STEP
        ang = ang.astype(np.float32)
        out = np.empty((n, angle_feat_size + 3), np.float32)
        out[:, :angle_feat_size] = get_angle_fts(ang[:, 0], ang[:, 1], angle_feat_size)
        out[:, angle_feat_size:] = dist          # (float64 -> float32 on assignment, as .astype did)
        return out

    def pair_dists(self, gmap_vpids, first=2):
        """symmetric [G, G] float32 of map distances between the real nodes gmap_vpids[first:] (M/r2r/agent.py:191-195)."""
        G = len(gmap_vpids)
        out = np.zeros((G, G), np.float32)
        if G > first:
            ix = np.array([self.graph._ix(vp) for vp in gmap_vpids[first:]], dtype=np.int64)
            sub = self.graph.D[np.ix_(ix, ix)].astype(np.float32)
            np.fill_diagonal(sub, 0)
            out[first:, first:] = sub
        return out


# ------------------------------------------------------------------------------------------------ node embeddings (device)
class NodeEmbedStore:
    """`node_embeds` of B GraphMaps (graph_utils.py:98,113-125) as index bookkeeping over a pool of device rows.

    The panorama encoder's outputs of every step are appended to the pool ([B*W_t] view rows, then [B] fused rows); a node is either
    ("set", row): rewritten by its own visit, or ("acc", [rows]): the running sum of the candidate views that saw it, read back as
    the mean.  `gather` turns the current state into the CSR index of goat_gather_segmean_* over the concatenated pool; its backward
    sends each map token's gradient to the rows it averaged (inverse index: one writer per pool row, no atomics)."""

    def __init__(self, B):
        self.B = B
        self.state = [dict() for _ in range(B)]
        self.pool, self.rows = [], 0
        self._view_base = self._fused_base = self._W = None

    def advance(self, B, W):
        """row bookkeeping of one step without tensors (host-side planning: TeacherEpisode.plan)."""
        self._view_base, self._W = self.rows, W
        self.rows += B * W
        self._fused_base = self.rows
        self.rows += B

    def begin_step(self, pano_embeds, fused):
        """register this step's panorama tokens [B, W, H] and fused / averaged panorama vectors [B, H]."""
        B, W, H = pano_embeds.shape
        self.advance(B, W)
        self.pool.append(pano_embeds.reshape(B * W, H))
        self.pool.append(fused.to(pano_embeds.dtype))

    def rewrite(self, b, vp):
        """update_node_embed(vp, avg_pano_embeds[b], rewrite=True)"""
        self.state[b][vp] = ('set', self._fused_base + b)

    def accumulate(self, b, vp, j):
        """update_node_embed(vp, pano_embeds[b, j])"""
        row = self._view_base + b * self._W + j
        cur = self.state[b].get(vp)
        if cur is None:
            self.state[b][vp] = ('acc', [row])
        elif cur[0] == 'set':                       # [embed, 1] += embed: a rewritten node that is accumulated onto again
            self.state[b][vp] = ('acc', [cur[1], row])
        else:
            cur[1].append(row)

    def csr(self, gmap_vpids, G, mem_rows=None):
        """-> (idx, start, scale) int32 / int32 / float32 numpy for output token (b, g) = segment b * G + g.  Slot 0 ([stop]) is
        empty (zeros); slot 1 ([MEM]) reads pool row mem_rows[b] if given; slots >= 2 the node's rows."""
        idx, start, scale = [], [0], []
        for b in range(self.B):
            vps, state = gmap_vpids[b], self.state[b]
            n = min(max(len(vps), 2), G)
            for g in range(n):
                rows, sc = (), 1.0
                if g == 1 and mem_rows is not None:
                    rows = (mem_rows[b],)
                elif g >= 2 and g < len(vps) and vps[g] is not None:
                    kind, r = state[vps[g]]
                    if kind == 'set':
                        rows = (r,)
                    else:
                        rows, sc = r, 1.0 / len(r)
                idx.extend(rows)
                scale.append(sc)
                start.append(len(idx))
            if G > n:                           # the padding slots of the bucket: empty segments
                scale.extend([1.0] * (G - n))
                start.extend([len(idx)] * (G - n))
        if not idx:
            idx = [-1]
        return np.asarray(idx, np.int32), np.asarray(start, np.int32), np.asarray(scale, np.float32)

    def gather(self, gmap_vpids, G, last_embeds=None):
        """gmap_img_embeds [B, G, H] (M/r2r/agent.py:180-185): zeros, the previous step's [MEM] state, then the node embeddings."""
        from . import graphmap, hipops
        pool = list(self.pool)
        mem_rows = None
        if last_embeds is not None:
            mem_rows = [self.rows + b for b in range(self.B)]
            pool.append(last_embeds.to(pool[0].dtype))
        src = torch.cat(pool, 0)
        idx, start, scale = self.csr(gmap_vpids, G, mem_rows)
        inv = graphmap.inverse_index(idx, start, scale, src.shape[0])
        dev = src.device
        out = hipops.gather_segmean(src, torch.from_numpy(idx).to(dev), torch.from_numpy(start).to(dev), torch.from_numpy(scale).to(dev),
                                    self.B * G, tuple(t.to(dev) for t in inv))
        return out.view(self.B, G, src.shape[1])


# ------------------------------------------------------------------------------------------------ the navigator
class ScanGraph:
    """One building: viewpoint ids, positions [N, 3] float64, undirected adjacency (M/utils/data.py:80-105 `load_nav_graphs`)."""

    def __init__(self, name, vpids, positions, edges):
        self.name, self.vpids = name, list(vpids)
        self.index = {v: i for i, v in enumerate(self.vpids)}
        self.pos = np.asarray(positions, np.float64).reshape(len(self.vpids), 3)
        self.adj = [[] for _ in self.vpids]
        for a, b in edges:
            ia, ib = self.index[a], self.index[b]
            if ib not in self.adj[ia]:
                self.adj[ia].append(ib)
                self.adj[ib].append(ia)
        self._cands = {}
        self._sp = None

    @staticmethod
    def from_connectivity(connectivity_dir, scan):
        """Matterport3D `<scan>_connectivity.json` (list of {image_id, pose[16], included, unobstructed[]})."""
        with open(os.path.join(connectivity_dir, '%s_connectivity.json' % scan)) as f:
            data = json.load(f)
        vpids, pos, edges = [], [], []
        for i, item in enumerate(data):
            if not item['included']:
                continue
            for j, conn in enumerate(item['unobstructed']):
                if conn and data[j]['included']:
                    if item['image_id'] not in vpids:
                        vpids.append(item['image_id'])
                        pos.append([item['pose'][3], item['pose'][7], item['pose'][11]])
                    edges.append((item['image_id'], data[j]['image_id']))
        keep = set(vpids)
        return ScanGraph(scan, vpids, pos, [(a, b) for a, b in edges if a in keep and b in keep])

    @staticmethod
    def synthetic(name='scan0', n=40, seed=0, degree=3, extent=12.0):
        """random planar-ish scan: points in a box, each joined to its `degree` nearest neighbours (connected by construction:
        node i > 0 is also joined to its nearest predecessor)."""
        rs = np.random.RandomState(seed)
        pos = np.concatenate([rs.uniform(-extent, extent, (n, 2)), rs.uniform(-1.5, 1.5, (n, 1))], 1)
        vpids = ['%s_vp%03d' % (name, i) for i in range(n)]
        d = np.sqrt(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1))
        np.fill_diagonal(d, np.inf)
        edges = set()
        for i in range(n):
            for j in np.argsort(d[i])[:degree]:
                edges.add((min(i, int(j)), max(i, int(j))))
            if i > 0:
                j = int(np.argmin(d[i, :i]))
                edges.add((j, i))
        return ScanGraph(name, vpids, pos, [(vpids[a], vpids[b]) for a, b in sorted(edges)])

    # all-pairs shortest distances / predecessor matrix (networkx all_pairs_dijkstra in M/r2r/env.py:183-189)
    def shortest(self):
        if self._sp is None:
            from scipy.sparse import csr_matrix
            from scipy.sparse.csgraph import dijkstra
            n = len(self.vpids)
            rows, cols, w = [], [], []
            for i in range(n):
                for j in self.adj[i]:
                    rows.append(i)
                    cols.append(j)
                    w.append(float(np.sqrt(((self.pos[i] - self.pos[j]) ** 2).sum())))
            dist, pred = dijkstra(csr_matrix((w, (rows, cols)), shape=(n, n)), directed=False, return_predecessors=True)
            self._sp = (dist, pred)
        return self._sp

    def shortest_path(self, a, b):
        dist, pred = self.shortest()
        i, j = self.index[a], self.index[b]
        out = [j]
        while out[-1] != i:
            out.append(int(pred[i, out[-1]]))
        return [self.vpids[k] for k in reversed(out)]

    def candidates(self, vp):
        """make_candidate (M/r2r/env.py:241-333) for viewpoint `vp`, base-view independent part: one entry per neighbour with its
        absolute heading / elevation ('normalized_*'), the view index that sees it closest to its centre ('pointId') and its
        position, in the order the 36-view sweep first meets them."""
        got = self._cands.get(vp)
        if got is not None:
            return got
        i = self.index[vp]
        nb = self.adj[i]
        out = []
        if nb:
            h, e, _ = rel_pos(self.pos[i], self.pos[nb])
            first = []
            for n_i, (hh, ee) in enumerate(zip(h, e)):
                best, best_d, first_ix, first_d = 0, float('inf'), None, None
                for ix in range(36):
                    vh, ve = view_angles(ix)
                    rh = (hh - vh + math.pi) % (2 * math.pi) - math.pi
                    re = ee - ve
                    dd = math.sqrt(rh * rh + re * re)
                    if dd < best_d:
                        best, best_d = ix, dd
                    if first_ix is None and abs(rh) < HFOV / 2 and abs(re) < VFOV / 2:
                        first_ix, first_d = ix, dd
                if first_ix is None:
                    first_ix, first_d = best, best_d
                first.append((first_ix, first_d, n_i))
                out.append({'viewpointId': self.vpids[nb[n_i]], 'pointId': best, 'normalized_heading': float(hh),
                            'normalized_elevation': float(ee), 'position': tuple(float(x) for x in self.pos[nb[n_i]]), 'scanId': self.name,
                            'idx': n_i + 1})
            out = [out[k] for _, _, k in sorted(first)]
        self._cands[vp] = out
        return out


class ObjectStore:
    """The object half of a REVERIE / SOON observation (M/reverie/data_utils.py:46-99 `ObjectFeatureDB`; M/reverie/env.py:451-479): per
    (scan, viewpoint) up to `max_objects` detected objects with an image feature, a viewing direction (heading, elevation), a bounding
    box size (w, h in pixels of the 640 x 480 frame), an object id and a category number (`obj_name` < 45).  The features of ALL objects
    live in one [sum O, D] table that is moved to the device once (`to`); observations carry ROW numbers, and `gather` assembles a
    batch's `reverie_obj_img_fts` with the kernel that gathers the view features (FeatureStore.gather) — the 768-wide rows are never
    touched on the host.  `attributes` is `get_object_feature`: angle features relative to the agent's heading / elevation, box
    features (h / 480, w / 640, their product)."""

    def __init__(self, entries, D, dtype=torch.bfloat16, max_objects=None):
        """entries: {'<scan>_<vp>': dict(fts float32 [O, >= D], directions [O, 2], sizes [O, 2] (w, h), obj_ids [O], names int [O])}"""
        self.start, self.count, self.attrs = {}, {}, {}
        blocks, n = [], 0
        for k, e in entries.items():
            o = len(e['obj_ids']) if max_objects is None else min(len(e['obj_ids']), max_objects)
            self.start[k], self.count[k] = n, o
            self.attrs[k] = {'directions': np.asarray(e['directions'], np.float64).reshape(-1, 2)[:o], 'sizes': np.asarray(e['sizes'], np.float64).reshape(-1, 2)[:o],
                             'obj_ids': list(e['obj_ids'])[:o], 'names': np.asarray(e['names'], np.int64)[:o]}
            blocks.append(np.asarray(e['fts'], np.float32).reshape(-1, np.asarray(e['fts']).shape[-1] if len(e['obj_ids']) else D)[:o, :D])
            n += o
        tab = np.concatenate(blocks, 0) if n else np.zeros((0, D), np.float32)
        from .features import FeatureStore
        self._fs = FeatureStore.__new__(FeatureStore)               # (row gather only: the table is [sum O, D], one row per object)
        self._fs.keys, self._fs.index, self._fs.views = [], {}, 1
        self._fs.table = torch.from_numpy(np.ascontiguousarray(tab)).to(dtype)
        self._fs.dev = None
        self.D = D

    @classmethod
    def from_hdf5(cls, path, D, category_of=None, max_objects=None, dtype=torch.bfloat16):
        """The reference's object feature file (M/reverie/data_utils.py:46-78, P/data/dataset.py:838-861): one dataset '<scan>_<viewpoint>'
        [O, >= D] per viewpoint with attributes `directions` [O, 2], `sizes` [O, 2], `obj_ids` [O], `names` [O] (and `bboxes`, unused
        here).  category_of: name string -> category number (the reference's `preprocess_name` over its category-mapping files, which
        are dataset files and not part of this path); None: the names must already be numbers.  Read through h5py or, without it,
        libhdf5 (h5lite)."""
        from . import h5lite
        entries = {}
        with h5lite.open_file(path, 'r') as f:
            for key in f.keys():
                ds = f[key]
                attrs = dict(ds.attrs.items())
                names = list(np.asarray(attrs.get('names', [])).reshape(-1))
                names = [category_of(n.decode() if isinstance(n, bytes) else str(n)) for n in names] if category_of is not None else [int(n) for n in names]
                ids = [i.decode() if isinstance(i, bytes) else (i if isinstance(i, str) else (int(i) if float(i).is_integer() else i))
                       for i in np.asarray(attrs.get('obj_ids', [])).reshape(-1)]
                entries[key] = {'fts': np.asarray(ds[...], np.float32), 'directions': attrs.get('directions', np.zeros((0, 2))),
                                'sizes': attrs.get('sizes', np.zeros((0, 2))), 'obj_ids': ids, 'names': names}
        return cls(entries, D, dtype, max_objects)

    @classmethod
    def synthetic(cls, scans, D=768, max_objects=20, seed=0, dtype=torch.bfloat16, p_empty=0.2):
        """random objects on every viewpoint of the given ScanGraphs (object ids unique per scan; ~p_empty of the viewpoints see none)."""
        rs = np.random.RandomState(seed)
        entries = {}
        for scan in scans:
            next_id = 0
            for vp in scan.vpids:
                o = 0 if rs.uniform() < p_empty else int(rs.randint(1, max_objects + 1))
                entries['%s_%s' % (scan.name, vp)] = {
                    'fts': rs.standard_normal((o, D)).astype(np.float32), 'directions': np.stack([rs.uniform(0, 2 * np.pi, o), rs.uniform(-0.5, 0.5, o)], 1),
                    'sizes': np.stack([rs.uniform(20, 640, o), rs.uniform(20, 480, o)], 1), 'obj_ids': list(range(next_id, next_id + o)),
                    'names': rs.randint(0, 45, o)}
                next_id += o
        return cls(entries, D, dtype)

    def to(self, device):
        self._fs.to(device)
        return self

    def meta(self):
        """a copy WITHOUT the feature table (row numbers, directions, sizes, ids, names only): what a host-side planner needs
        (PlanWorker ships it to its worker process; `attributes` works, `gather` / `table` do not)."""
        m = ObjectStore.__new__(ObjectStore)
        m.start, m.count, m.attrs, m.D, m._fs = self.start, self.count, self.attrs, self.D, None
        return m

    @property
    def table(self):
        return self._fs.table

    def attributes(self, scan, vp, base_heading, base_elevation, angle_feat_size=4):
        """-> (rows int64 [O], obj_ang_fts [O, angle_feat_size], obj_box_fts [O, 3], obj_ids, obj_names) — get_object_feature (:80-99)."""
        k = '%s_%s' % (scan, vp)
        o, a = self.count.get(k, 0), self.attrs.get(k)
        ang = np.zeros((o, angle_feat_size), np.float32)
        box = np.zeros((o, 3), np.float32)
        if o:
            for j in range(o):
                ang[j] = angle_feature(a['directions'][j, 0] - base_heading, a['directions'][j, 1] - base_elevation, angle_feat_size)
                w, h = a['sizes'][j]
                box[j, :2] = [h / 480, w / 640]
                box[j, 2] = box[j, 0] * box[j, 1]
        rows = np.arange(self.start.get(k, 0), self.start.get(k, 0) + o, dtype=np.int64)
        return rows, ang, box, (a['obj_ids'] if o else []), (a['names'] if o else np.zeros(0, np.int64))

    def gather(self, obj_rows, out_dtype=None):
        """obj_rows int64 [B, O] (-1 = padding) on the table's device -> [B, O, D]"""
        return self._fs.gather(obj_rows, out_dtype)

    def host_rows(self, obj_rows):
        return self._fs.host_rows(obj_rows)


class GraphSim:
    """The batch of simulators of EnvBatch + R2RNavBatch._get_obs (M/r2r/env.py:26-96,335-377) on ScanGraphs.

    episodes: list of dicts {instr_id, scan (ScanGraph), path [vpids], heading, instr_encoding}.  `features`: an object with
    `row(scan_name, vpid) -> int` (features.FeatureStore): observations carry feature ROW numbers, not feature arrays."""

    def __init__(self, features=None, angle_feat_size=4, objects=None, seed=0, obj_fallback=True):
        self.features = features
        self.obj_rng = np.random.RandomState(seed)      # draws the stand-in target object of episodes without one (see _gt_obj_id)
        self.obj_fallback = obj_fallback                # False: such episodes keep gt_obj_id None (fixtures generated without the random draw)
        self.objects = objects              # ObjectStore: REVERIE / SOON observations (M/reverie/env.py:451-486); episodes then carry
        self.angle_feat_size = angle_feat_size      # 'obj_id' (the target object, may be None) and 'end_vps' (viewpoints that see it)
        self.view_angle_fts = view_angle_feature_table(angle_feat_size)
        self.batch, self.state = [], []

    def _gt_obj_id(self, ep, obj_ids):
        """target object of an observation (M/reverie/env.py:481-484): the episode's `objId`; an episode WITHOUT one (the augmented
        data) gets a random object of the current viewpoint — np.random.choice(obj_ids) there, this simulator's seeded generator here —
        so that such episodes contribute to the object-grounding loss as in the reference; None only when the viewpoint has no objects."""
        if ep.get('obj_id') is not None or len(obj_ids) == 0 or not self.obj_fallback:
            return ep.get('obj_id')
        return obj_ids[int(self.obj_rng.randint(len(obj_ids)))]

    @staticmethod
    def _snap(heading, elevation):
        """discretised viewing angles: heading / elevation snapped to the 30-degree grid, view index 0..35"""
        hs = int(round(heading / math.radians(30))) % 12
        es = min(2, max(0, int(round(elevation / math.radians(30))) + 1))
        return es * 12 + hs

    def reset(self, episodes):
        self.batch = list(episodes)
        self.state = [(ep['path'][0], self._snap(ep['heading'], 0.0)) for ep in self.batch]
        return self.observe()

    def step(self, moves):
        """moves[i] = (viewpoint, view index) or None (stay): M/r2r/agent.py:349-378 teleports with newEpisode."""
        for i, mv in enumerate(moves):
            if mv is not None:
                self.state[i] = mv
        return self.observe()

    def observe(self):
        obs = []
        for ep, (vp, view) in zip(self.batch, self.state):
            scan = ep['scan']
            bh, be = view_angles(view)
            cands = []
            for c in scan.candidates(vp):
                c = dict(c)
                c['heading'] = c['normalized_heading'] - bh
                c['elevation'] = c['normalized_elevation'] - be
                c['angle_feat'] = angle_feature(c['heading'], c['elevation'], self.angle_feat_size)
                cands.append(c)
            dist, _ = scan.shortest()
            obs.append({'instr_id': ep['instr_id'], 'scan': scan.name, 'scan_graph': scan, 'viewpoint': vp, 'viewIndex': view,
                        'position': tuple(float(x) for x in scan.pos[scan.index[vp]]), 'heading': bh, 'elevation': be,
                        'feature_row': self.features.row(scan.name, vp) if self.features is not None else -1,
                        'view_angle_fts': self.view_angle_fts[view], 'candidate': cands,
                        'instr_encoding': ep['instr_encoding'], 'gt_path': ep['path'],
                        'distance': float(dist[scan.index[vp], scan.index[ep['path'][-1]]])})
            if self.objects is not None:
                # (the simulator reports the continuous heading; on this navigator it is the snapped view's, as for the candidates)
                rows, ang, box, ids, names = self.objects.attributes(scan.name, vp, bh, be, self.angle_feat_size)
                ob = obs[-1]
                ob.update({'obj_rows': rows, 'obj_ang_fts': ang, 'obj_box_fts': box, 'obj_ids': ids, 'obj_name': names,
                           'gt_end_vps': ep.get('end_vps', []), 'gt_obj_id': self._gt_obj_id(ep, ids)})
                if ep.get('end_vps'):           # several goal viewpoints on REVERIE: distance to the nearest (env.py:493-503)
                    ob['distance'] = float(min(dist[scan.index[vp], scan.index[e]] for e in ep['end_vps']))
        return obs


# ------------------------------------------------------------------------------------------------ input builders (agent.py)
def language_inputs(obs, pad_id=0):
    """_language_variable (M/r2r/agent.py:38-65) without the dictionaries (the caller adds the BACL / FACL tensors)."""
    lens = [len(ob['instr_encoding']) for ob in obs]
    ids = np.full((len(obs), max(lens)), pad_id, np.int64)
    mask = np.zeros((len(obs), max(lens)), bool)
    for i, ob in enumerate(obs):
        ids[i, :lens[i]] = ob['instr_encoding']
        mask[i, :lens[i]] = True
    return {'txt_ids': torch.from_numpy(ids), 'txt_masks': torch.from_numpy(mask)}


def panorama_inputs(obs, angle_feat_size=4, width=None, obj_width=None):
    """_panorama_feature_variable_do (M/r2r/agent.py:82-148): candidate views first (nav type 1), then the views no candidate
    used (nav type 0), padded to the longest panorama of the batch (or `width`).  The 768-wide image features are NOT assembled
    here: `view_rows[b, j]` = feature_row * 36 + view index of token j (-1: padding) for a device gather."""
    B = len(obs)
    rows, locs, types, cand_vpids, lens = [], [], [], [], []
    for ob in obs:
        used, r, ang, ty, cv = set(), [], [], [], []
        for cc in ob['candidate']:
            r.append(ob['feature_row'] * 36 + cc['pointId'])
            ang.append(cc['angle_feat'])
            ty.append(1)
            cv.append(cc['viewpointId'])
            used.add(cc['pointId'])
        rest = [k for k in range(36) if k not in used]
        base = ob['feature_row'] * 36
        r.extend(base + k for k in rest)
        ty.extend([0] * len(rest))
        loc = np.ones((len(r), angle_feat_size + 3), np.float32)
        if ang:
            loc[:len(ang), :angle_feat_size] = np.stack(ang, 0)
        loc[len(ang):, :angle_feat_size] = np.asarray(ob['view_angle_fts'])[rest]
        locs.append(loc)
        rows.append(r)
        types.append(ty)
        cand_vpids.append(cv)
        lens.append(len(r))
    W = max(lens) if width is None else width
    if W < max(lens):
        raise ValueError('panorama_inputs: a panorama has %d tokens, the bucket holds %d' % (max(lens), W))
    view_rows = np.full((B, W), -1, np.int64)
    loc_fts = np.zeros((B, W, angle_feat_size + 3), np.float32)
    nav_types = np.zeros((B, W), np.int64)
    for b in range(B):
        view_rows[b, :lens[b]] = rows[b]
        loc_fts[b, :lens[b]] = locs[b]
        nav_types[b, :lens[b]] = types[b]
    out = {'view_rows': torch.from_numpy(view_rows), 'loc_fts': torch.from_numpy(loc_fts), 'nav_types': torch.from_numpy(nav_types),
           'view_lens': torch.tensor(lens, dtype=torch.int64), 'cand_vpids': cand_vpids}
    if 'obj_ids' in obs[0]:
        out.update(panorama_object_inputs(obs, out, angle_feat_size, obj_width))
    return out


def panorama_object_inputs(obs, pano, angle_feat_size=4, obj_width=None):
    """The object half of `_panorama_feature_variable_do` of the REVERIE agent (M/reverie/agent_obj_goat.py:180-271): object tokens follow
    the views of their panorama (nav type 2), `loc_fts` / `nav_types` cover views + objects (padded to the longest row of the batch),
    `reverie_obj_*` are the per-object tensors the image embedding consumes.  obj_rows[b, j] = row of ObjectStore.table (-1: padding)."""
    B = len(obs)
    A = angle_feat_size + 3
    olens = [len(ob['obj_ids']) for ob in obs]
    vlens = [int(x) for x in pano['view_lens']]
    O = max(olens) if obj_width is None else obj_width
    if O < max(olens):
        raise ValueError('panorama_object_inputs: a viewpoint has %d objects, the bucket holds %d' % (max(olens), O))
    Wv = pano['view_rows'].shape[1]
    W = max(v + o for v, o in zip(vlens, olens)) if obj_width is None else Wv + O
    obj_rows = np.full((B, O), -1, np.int64)
    obj_locs = np.zeros((B, O, A), np.float32)
    obj_names = np.zeros((B, O), np.int64)
    loc_fts = np.zeros((B, W, A), np.float32)
    nav_types = np.zeros((B, W), np.int64)
    rnav = np.zeros((B, 36 + O), np.int64)
    vloc, vty = pano['loc_fts'].numpy(), pano['nav_types'].numpy()
    for b, ob in enumerate(obs):
        v, o = vlens[b], olens[b]
        loc_fts[b, :v], nav_types[b, :v] = vloc[b, :v], vty[b, :v]
        if o:
            ol = np.concatenate([ob['obj_ang_fts'], ob['obj_box_fts']], 1)
            obj_rows[b, :o], obj_locs[b, :o], obj_names[b, :o] = ob['obj_rows'], ol, np.asarray(ob['obj_name'], np.int64)
            loc_fts[b, v:v + o], nav_types[b, v:v + o] = ol, 2
            rnav[b, 36:36 + o] = 2
    return {'loc_fts': torch.from_numpy(loc_fts), 'nav_types': torch.from_numpy(nav_types), 'obj_rows': torch.from_numpy(obj_rows),
            'reverie_obj_lens': torch.tensor(olens, dtype=torch.int64), 'reverie_obj_locs': torch.from_numpy(obj_locs),
            'reverie_obj_names': torch.from_numpy(obj_names), 'reverie_obj_nav_types': torch.from_numpy(rnav),
            'obj_ids': [list(ob['obj_ids']) for ob in obs]}


def gmap_order(gmap):
    """[stop], [MEM], visited nodes, unvisited nodes in the insertion order of node_positions (M/r2r/agent.py:159-176,
    enc_full_graph)."""
    visited = [k for k in gmap.node_positions if gmap.graph.visited(k)]
    unvisited = [k for k in gmap.node_positions if not gmap.graph.visited(k)]
    return [None, None] + visited + unvisited, [0, 1] + [1] * len(visited) + [0] * len(unvisited), len(unvisited) == 0


def gmap_inputs(obs, gmaps, width=None, angle_feat_size=4, mem_selectable=False):
    """_nav_gmap_variable (M/r2r/agent.py:151-237) without the embeddings (NodeEmbedStore.gather)."""
    B = len(obs)
    vpids, vis, no_left = zip(*[gmap_order(g) for g in gmaps])
    lens = [len(v) for v in vpids]
    G = max(lens) if width is None else width
    if G < max(lens):
        raise ValueError('gmap_inputs: a map has %d nodes, the bucket holds %d' % (max(lens), G))
    step_ids = np.zeros((B, G), np.int64)
    pos = np.zeros((B, G, angle_feat_size + 3), np.float32)
    pair = np.zeros((B, G, G), np.float32)
    vmask = np.zeros((B, G), bool)
    gmask = np.zeros((B, G), bool)
    for b, (ob, g) in enumerate(zip(obs, gmaps)):
        n = lens[b]
        step_ids[b, :n] = [g.node_step_ids.get(vp, 0) for vp in vpids[b]]
        pos[b, :n] = g.get_pos_fts(ob['viewpoint'], vpids[b], ob['heading'], ob['elevation'], angle_feat_size)
        pair[b, :n, :n] = g.pair_dists(vpids[b])
        vmask[b, :n] = np.asarray(vis[b], bool)
        gmask[b, :n] = True
    if not mem_selectable:              # the [MEM] token cannot be chosen (M/r2r/agent.py:209).  The REVERIE agent's copy of this builder lacks
        gmask[:, 1] = False             # that line (M/reverie/agent_obj_goat.py:273-343): there the slot stays selectable (its viewpoint id is
                                        # None: choosing it ends the episode like [stop]) — mem_selectable=True reproduces it
    return {'gmap_vpids': [list(v) for v in vpids], 'gmap_step_ids': torch.from_numpy(step_ids), 'gmap_pos_fts': torch.from_numpy(pos),
            'gmap_visited_masks': torch.from_numpy(vmask), 'gmap_pair_dists': torch.from_numpy(pair), 'gmap_masks': torch.from_numpy(gmask),
            'gmap_lens': lens, 'no_vp_left': list(no_left)}


def vp_inputs(obs, gmaps, cand_vpids, view_lens, nav_types, width, angle_feat_size=4, gmap_pos=None, obj_lens=None):
    """_nav_vp_variable_mem (M/r2r/agent.py:271-304) without the embeddings: [stop], [MEM], then the panorama tokens.
    width = panorama width + 2.  gmap_pos = (gmap_vpids, gmap_pos_fts [B, G, angle_feat_size + 3]) of gmap_inputs on the SAME
    observations: the candidates and the start node are nodes of the map and their features are seen from the same viewpoint under
    the same heading — the rows are taken from there instead of being computed a second and third time."""
    B = len(obs)
    A = angle_feat_size + 3
    pos = np.zeros((B, width, 2 * A), np.float32)
    for b, (ob, g) in enumerate(zip(obs, gmaps)):
        if gmap_pos is not None:
            where = {vp: j for j, vp in enumerate(gmap_pos[0][b]) if vp is not None}
            rows = gmap_pos[1][b]
            cand = rows[[where[vp] for vp in cand_vpids[b]]] if cand_vpids[b] else np.zeros((0, A), np.float32)
            start = rows[where[g.start_vp]]
        else:
            cand = g.get_pos_fts(ob['viewpoint'], cand_vpids[b], ob['heading'], ob['elevation'], angle_feat_size) if cand_vpids[b] else \
                np.zeros((0, A), np.float32)
            start = g.get_pos_fts(ob['viewpoint'], [g.start_vp], ob['heading'], ob['elevation'], angle_feat_size)
        pos[b, :, :A] = start
        pos[b, 2:len(cand) + 2, A:] = cand
    nav_types = torch.as_tensor(nav_types)
    view_lens = torch.as_tensor(view_lens)
    nav = torch.cat([torch.ones(B, 1, dtype=torch.bool), torch.zeros(B, 1, dtype=torch.bool), nav_types == 1], 1)
    if obj_lens is not None:        # REVERIE (_nav_vp_variable_do, M/reverie/agent_obj_goat.py:345-388): object tokens behind the views
        obj_lens = torch.as_tensor(obj_lens)
        masks = torch.arange(width)[None, :] < (view_lens + obj_lens + 2)[:, None]
        obj = torch.cat([torch.ones(B, 1, dtype=torch.bool), torch.zeros(B, 1, dtype=torch.bool), nav_types == 2], 1)
        return {'vp_pos_fts': torch.from_numpy(pos), 'vp_masks': masks, 'vp_nav_masks': nav, 'vp_obj_masks': obj,
                'vp_cand_vpids': [[None, None] + list(x) for x in cand_vpids]}
    masks = torch.arange(width)[None, :] < (view_lens + 2)[:, None]
    return {'vp_pos_fts': torch.from_numpy(pos), 'vp_masks': masks, 'vp_nav_masks': nav,
            'vp_cand_vpids': [[None, None] + list(x) for x in cand_vpids]}


def teacher_object(obs, ended, view_lens, ignoreid=-100):
    """_teacher_object (M/reverie/agent_obj_goat.py:419-436): at a goal viewpoint the index of the target object among the local tokens
    ([stop], [MEM], views, objects); everywhere else — and when the target is not among the detected objects — the ignore value."""
    t = np.full(len(obs), ignoreid, np.int64)
    for i, ob in enumerate(obs):
        if ended[i] or ob['viewpoint'] not in ob['gt_end_vps']:
            continue
        for j, oid in enumerate(ob['obj_ids']):
            if str(oid) == str(ob['gt_obj_id']):
                t[i] = j + int(view_lens[i]) + 2
                break
    return t


def teacher_action(obs, vpids, ended, visited_masks=None, imitation_learning=False, t=None, ignoreid=-100):
    """_teacher_action (M/r2r/agent.py:306-347), expert policy 'spl'."""
    a = np.zeros(len(obs), dtype=np.int64)
    for i, ob in enumerate(obs):
        if ended[i]:
            a[i] = ignoreid
        elif imitation_learning:
            assert ob['viewpoint'] == ob['gt_path'][t]
            if t == len(ob['gt_path']) - 1:
                a[i] = 0
            else:
                goal = ob['gt_path'][t + 1]
                for j, vpid in enumerate(vpids[i]):
                    if goal == vpid:
                        a[i] = j
                        break
        elif ob['viewpoint'] == ob['gt_path'][-1]:
            a[i] = 0
        else:
            scan = ob['scan_graph']
            dist, _ = scan.shortest()
            cur, goal = scan.index[ob['viewpoint']], scan.index[ob['gt_path'][-1]]
            best, best_d = ignoreid, float('inf')
            for j, vpid in enumerate(vpids[i]):
                if j > 1 and ((visited_masks is None) or (not visited_masks[i][j])):
                    k = scan.index[vpid]
                    d = dist[k, goal] + dist[cur, k]
                    if d < best_d:
                        best_d, best = d, j
            a[i] = best
    return a


# ------------------------------------------------------------------------------------------------ the rollout (agent.py:448-676)
class NavRollout:
    """One rollout of B episodes through a VLNBert-compatible `model(mode, batch)` (nav_model.VLNBert).

        sim = GraphSim(store); ro = NavRollout(model, sim, store, max_action_len=15)
        loss, traj = ro.run(episodes, feedback='teacher', extras={'language': {...BACL/FACL tensors...}, 'panorama': {...}, 'navigation': {...}})

    extras: tensors added verbatim to the input dict of the given mode (the confounder dictionaries z_dicts / z_front_dict of
    M/r2r/agent.py:486-511).  `pano_width` / `gmap_buckets`: pad every step's panorama to a fixed width and the map to the next
    bucket (shape-stable steps: a captured step graph per bucket can be replayed; None = the reference's per-batch maxima)."""

    def __init__(self, model, sim, features, max_action_len=15, fusion='dynamic', ignoreid=-100, pano_width=None, gmap_buckets=None,
                 device='cuda', hoist_text_kv=True, obj_width=None, teacher_scores=True):
        self.model, self.sim, self.features = model, sim, features
        self.teacher_scores = teacher_scores    # teacher forcing records stop scores / best objects too, as the reference (one read-back per step)
        self.objects = getattr(sim, 'objects', None)      # REVERIE / SOON: object tokens + object grounding (M/reverie/agent_obj_goat.py:560-790)
        self.obj_width = obj_width
        self.max_action_len, self.fusion, self.ignoreid = max_action_len, fusion, ignoreid
        self.pano_width, self.gmap_buckets = pano_width, gmap_buckets
        self.hoist_text_kv = hoist_text_kv      # K|V projections of the instruction once per episode (nav_model.text_kv) instead of per step
        self.device = torch.device(device)
        self.host_s = 0.0          # seconds spent in the host-side builders during the last run (diagnostics)

    def _bucket(self, n):
        if not self.gmap_buckets:
            return None
        for g in self.gmap_buckets:
            if g >= n:
                return g
        raise ValueError('map with %d nodes exceeds the largest bucket %d' % (n, self.gmap_buckets[-1]))

    def build_step(self, obs, gmaps, t, ended, imitation):
        """every host-built table of step t (no device work, no dependence on model outputs)."""
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.node_step_ids[obs[i]['viewpoint']] = t + 1
        pano = panorama_inputs(obs, self.sim.angle_feat_size, self.pano_width, self.obj_width)
        return pano

    def run(self, episodes, feedback='teacher', extras=None, train_ml=1.0, compute_loss=True, sampler=None):
        import time
        from collections import defaultdict
        dd = lambda d: defaultdict(lambda: None, d)
        extras = extras or {}
        dev = self.device
        mv = lambda d: {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in d.items()}
        t_host = time.perf_counter()
        obs = self.sim.reset(episodes)
        B = len(obs)
        gmaps = [GraphMap(ob['viewpoint']) for ob in obs]
        for g, ob in zip(gmaps, obs):
            g.update_graph(ob)
        traj = [{'instr_id': ob['instr_id'], 'path': [[ob['viewpoint']]]} for ob in obs]
        if self.objects is not None:
            for tr in traj:
                tr['pred_objid'] = None
        lang = language_inputs(obs)
        self.host_s = time.perf_counter() - t_host
        lang_in = mv(lang)
        lang_in.update(extras.get('language', {}))
        txt_embeds = self.model('language', dd(lang_in))
        txt_kv = self.model('text_kv', {'txt_embeds': txt_embeds}) if self.hoist_text_kv else None
        store = NodeEmbedStore(B)
        ended = np.zeros(B, bool)
        just_ended = np.zeros(B, bool)
        last_embeds = None
        ml_loss = 0.0
        og_loss = 0.0
        has_obj = self.objects is not None
        steps = 0
        self.actions = []           # the action index of every sample at every step taken (TeacherEpisode.plan(actions=) re-walks them)
        for t in range(self.max_action_len):
            t_host = time.perf_counter()
            pano = self.build_step(obs, gmaps, t, ended, feedback == 'teacher')
            self.host_s += time.perf_counter() - t_host
            pin = {'view_img_fts': self.features.gather(pano['view_rows'].to(dev, non_blocking=True)), 'loc_fts': pano['loc_fts'].to(dev, non_blocking=True),
                   'nav_types': pano['nav_types'].to(dev, non_blocking=True), 'view_lens': pano['view_lens'].to(dev, non_blocking=True),
                   'already_dropout': False}
            if has_obj:
                if pano['obj_rows'].shape[1] == 0:          # no viewpoint of this step sees an object: one padding slot (the kernels take no
                    pano['obj_rows'] = torch.full((B, 1), -1, dtype=torch.int64)                # zero-width tensors; obj_lens = 0 masks it)
                    pano['reverie_obj_names'] = torch.zeros((B, 1), dtype=torch.int64)
                    pano['reverie_obj_locs'] = torch.zeros((B, 1, pano['loc_fts'].shape[2]), dtype=torch.float32)
                pin.update({'reverie_obj_img_fts': self.objects.gather(pano['obj_rows'].to(dev, non_blocking=True)),
                            'reverie_obj_lens': pano['reverie_obj_lens'].to(dev, non_blocking=True),
                            'reverie_obj_names': pano['reverie_obj_names'].to(dev, non_blocking=True),
                            'reverie_obj_locs': pano['reverie_obj_locs'].to(dev, non_blocking=True),
                            'reverie_obj_nav_types': pano['reverie_obj_nav_types'].to(dev, non_blocking=True)})
            pin.update(extras.get('panorama', {}))
            pano_embeds, pano_masks, fused = self.model('panorama', dd(pin))
            if fused is None:                                   # not adaptive_pano_fusion: masked mean (M/r2r/agent.py:545-547)
                fused = torch.sum(pano_embeds * pano_masks.unsqueeze(2), 1) / torch.sum(pano_masks, 1, keepdim=True)
            t_host = time.perf_counter()
            store.begin_step(pano_embeds, fused)
            for i, g in enumerate(gmaps):
                if not ended[i]:
                    store.rewrite(i, obs[i]['viewpoint'])
                    for j, cvp in enumerate(pano['cand_vpids'][i]):
                        if not g.graph.visited(cvp):
                            store.accumulate(i, cvp, j)
            n_nodes = max(2 + len(g.node_positions) for g in gmaps)
            gin = gmap_inputs(obs, gmaps, self._bucket(n_nodes), self.sim.angle_feat_size, mem_selectable=has_obj)
            W = pano['nav_types'].shape[1]              # (REVERIE: views + objects)
            vin = vp_inputs(obs, gmaps, pano['cand_vpids'], pano['view_lens'], pano['nav_types'], W + 2, self.sim.angle_feat_size,
                            gmap_pos=(gin['gmap_vpids'], gin['gmap_pos_fts'].numpy()), obj_lens=pano['reverie_obj_lens'] if has_obj else None)
            G = gin['gmap_step_ids'].shape[1]
            nav_vpids = gin['gmap_vpids'] if self.fusion != 'local' else vin['vp_cand_vpids']
            target = None
            if compute_loss or feedback == 'teacher':
                target = teacher_action(obs, nav_vpids, ended, visited_masks=gin['gmap_visited_masks'].numpy() if self.fusion != 'local' else None,
                                        imitation_learning=(feedback == 'teacher' and not has_obj), t=t, ignoreid=self.ignoreid)      # (the REVERIE agent has the shortest-path expert only, M/reverie/agent_obj_goat.py:390-417)
            self.host_s += time.perf_counter() - t_host
            zero = pano_embeds.new_zeros(B, 1, pano_embeds.shape[-1])
            memtok = zero if last_embeds is None else last_embeds.unsqueeze(1).to(pano_embeds.dtype)
            nin = {'txt_embeds': txt_embeds, 'txt_masks': lang_in['txt_masks'],
                   'gmap_img_embeds': store.gather(gin['gmap_vpids'], G, last_embeds),
                   'vp_img_embeds': torch.cat([zero, memtok, pano_embeds], 1), 'vp_obj_masks': None, 'flops_count': False, 'txt_kv': txt_kv}
            nin.update(mv({k: v for k, v in gin.items() if k not in ('gmap_lens', 'no_vp_left')}))
            nin.update(mv(vin))
            nin.update(extras.get('navigation', {}))
            out = self.model('navigation', dd(nin))
            last_embeds = out['cls_embeds']
            logits = {'local': out['local_logits'], 'global': out['global_logits']}.get(self.fusion, out['fused_logits'])
            steps += 1
            if target is not None and compute_loss:
                ml_loss = ml_loss + torch.nn.functional.cross_entropy(logits.float(), torch.from_numpy(target).to(dev, non_blocking=True),
                                                                      reduction='sum', ignore_index=self.ignoreid)
                if has_obj:                 # object grounding at the goal viewpoints (M/reverie/agent_obj_goat.py:705-707)
                    otgt = teacher_object(obs, ended, pano['view_lens'], self.ignoreid)
                    og_loss = og_loss + torch.nn.functional.cross_entropy(out['obj_logits'].float(), torch.from_numpy(otgt).to(dev, non_blocking=True),
                                                                          reduction='sum', ignore_index=self.ignoreid)
            if feedback not in ('teacher', 'argmax', 'sample'):
                raise ValueError('invalid feedback option %r' % (feedback,))
            if feedback == 'teacher' and not self.teacher_scores:
                a_t = target
                stop = [ob['viewpoint'] == ob['gt_path'][-1] for ob in obs]
            else:
                # ONE device -> host copy per step: the chosen actions and the stop probabilities (M/r2r/agent.py:575-580,601-607).  The
                # reference records the stop score (and the best object) of the current node in EVERY feedback mode
                # (M/reverie/agent_obj_goat.py:679-689,754-761): teacher forcing too, so that its trajectories end with the same stop-node
                # backtrack and predicted object (`teacher_scores=False` skips the read-back where only the loss is used)
                probs = torch.softmax(logits.detach().float(), 1)
                if feedback == 'teacher':
                    act = torch.from_numpy(np.asarray(target, np.int64)).to(probs.device)
                elif feedback == 'argmax':
                    act = probs.argmax(1)
                elif sampler is not None:           # (a fixed action sequence in place of Categorical.sample(): the sampled-rollout golden)
                    act = torch.as_tensor(sampler(t, probs), dtype=torch.int64, device=probs.device)
                else:
                    act = torch.distributions.Categorical(probs).sample()
                rows = [act.to(torch.float32), probs[:, 0].detach()]
                if has_obj:                 # the best object of every sample's current viewpoint rides in the same read-back (:680-690)
                    vl = pano['view_lens'].to(dev)
                    pos = torch.arange(out['obj_logits'].shape[1], device=dev)[None, :]
                    ol = torch.where(pos >= (vl + 2)[:, None], out['obj_logits'].detach().float(), torch.full_like(out['obj_logits'], -float('inf'), dtype=torch.float32))
                    rows.append((ol.argmax(1) - (vl + 2)).to(torch.float32))
                back = torch.stack(rows, 0).cpu().numpy()
                a_t = target if feedback == 'teacher' else back[0].astype(np.int64)
                stop = (a_t == 0) if feedback == 'argmax' else [ob['viewpoint'] == ob['gt_path'][-1] for ob in obs]
                for i, g in enumerate(gmaps):
                    if not ended[i]:
                        g.node_stop_scores[obs[i]['viewpoint']] = {'stop': float(back[1, i])}
                        if has_obj:
                            ids = obs[i]['obj_ids']
                            g.node_stop_scores[obs[i]['viewpoint']]['og'] = ids[int(back[2, i])] if len(ids) > 0 else None
            t_host = time.perf_counter()
            self.actions.append(np.where(ended, 0, np.asarray(a_t, np.int64)))
            moves = []
            for i in range(B):
                forced = bool(stop[i] or ended[i] or gin['no_vp_left'][i] or t == self.max_action_len - 1)
                nxt = None if forced else nav_vpids[i][int(a_t[i])]
                if forced:
                    just_ended[i] = True            # M/r2r/agent.py:657-660: ONLY these four conditions mark the episode for the stop-node backtrack;
                if nxt is None:                     # a sampled action 0 (the [stop] token: nav_vpids[i][0] is None, :661) ends it without one
                    moves.append(None)
                else:
                    hop = gmaps[i].graph.path(obs[i]['viewpoint'], nxt)
                    traj[i]['path'].append(hop)
                    prev = traj[i]['path'][-2][-1] if len(hop) == 1 else hop[-2]
                    view = next(c['pointId'] for c in obs[i]['scan_graph'].candidates(prev) if c['viewpointId'] == nxt)
                    moves.append((nxt, view))
            # go back to the node with the best stop score (M/r2r/agent.py:665-672), in every feedback mode as the reference (with
            # teacher_scores=False a teacher rollout records no scores: its dictionary is empty and nothing moves)
            if True:
                for i in range(B):
                    if (not ended[i]) and just_ended[i] and gmaps[i].node_stop_scores:
                        stop_node, score = max(gmaps[i].node_stop_scores.items(), key=lambda kv: kv[1]['stop'])
                        if obs[i]['viewpoint'] != stop_node:
                            traj[i]['path'].append(gmaps[i].graph.path(obs[i]['viewpoint'], stop_node))
                        if has_obj:
                            traj[i]['pred_objid'] = score.get('og')             # (:761)
            obs = self.sim.step(moves)
            for i, ob in enumerate(obs):
                if not ended[i]:
                    gmaps[i].update_graph(ob)
            ended = np.logical_or(ended, np.array([m is None for m in moves]))
            self.host_s += time.perf_counter() - t_host
            if ended.all():
                break
        loss = ml_loss * train_ml / B if compute_loss else None
        # (the two parts are kept as DETACHED values: an attribute holding a tensor of the autograd graph would keep the graph — and its
        #  AccumulateGrad nodes, bound to this pass's stream — alive past the caller's backward; a later capture of a training step then
        #  accumulates off-graph, see hipops.graph)
        self.ml_loss = loss.detach() if torch.is_tensor(loss) else loss
        self.og_loss = None
        if compute_loss and has_obj:            # self.loss += ml_loss; self.loss += og_loss, both * train_ml / batch_size (:781-787)
            og = og_loss * train_ml / B
            self.og_loss = og.detach()
            loss = loss + og
        self.steps = steps
        return loss, traj


# ------------------------------------------------------------------------------------------------ teacher-forced episodes as ONE graph
def default_gmap_width(t, max_degree=7, granule=16):
    """map bucket of step t: [stop], [MEM], t + 1 visited nodes and at most max_degree fresh candidates per visit, rounded up."""
    n = 2 + (t + 1) * (1 + max_degree)
    return (n + granule - 1) // granule * granule


class TeacherEpisode:
    """A teacher-forced rollout (imitation learning, feedback = 'teacher': M/r2r/agent.py:414-420,592-594) as shape-stable device
    work.  With teacher forcing nothing the host builds depends on a model output: the walk follows the ground-truth paths, so the
    maps, positions, masks, logit-fusion matrices, targets and the gather indices of the node embeddings of ALL steps are known
    before the first kernel runs.  `plan()` builds them (numpy, fixed shapes: text bucket L, panorama width W, map width per
    step), `EpisodeBuffers` holds them in one device buffer fed by one pinned H2D copy, and `body()` is pure device code over
    those tensors — captured once into a hipGraph, it is replayed for every new batch of episodes.  Episodes shorter than the
    plan's step count carry target -100 (ignored) after their end, as the reference's `ended` bookkeeping does."""

    def __init__(self, sim, features, n_steps, text_len, pano_width=40, gmap_width=default_gmap_width, fusion='dynamic', ignoreid=-100,
                 obj_width=20):
        self.sim, self.features = sim, features
        self.T, self.L, self.W, self.gw = n_steps, text_len, pano_width, gmap_width
        self.fusion, self.ignoreid = fusion, ignoreid
        # REVERIE / SOON (sim.objects: an ObjectStore): up to `obj_width` object tokens behind the views of every panorama, the
        # object-grounding loss at goal viewpoints (M/reverie/agent_obj_goat.py:560-790).  WT = tokens per panorama (views + objects).
        self.objects = getattr(sim, 'objects', None)
        self.O = obj_width if self.objects is not None else 0
        self.WT = self.W + self.O

    # ---- host ---------------------------------------------------------------------------------------------------------------
    def plan(self, episodes, actions=None):
        """host tables of all T steps.  actions=None: the teacher-forced walk along the ground-truth paths (imitation labels).
        actions = [steps, B] action indices into the step's map nodes (0 = [stop]), e.g. NavRollout.actions of a sampled rollout run
        under no_grad: the walk follows THEM — the policy's own path — and the targets are the DAgger labels of the states it visits
        (`teacher_action(imitation_learning=False)`: M/r2r/agent.py:325-347), the episode ends as the sampled rollout does
        (:601-607,657-663: at the goal, on a sampled [stop], with no node left, at the last step).  One replay of the captured body then
        gives the loss and the gradients of the sampled half of the dagger iteration (:436-437) without an eager autograd pass."""
        if actions is not None:
            actions = [np.asarray(a, np.int64) for a in actions]
            if len(actions) > self.T:
                raise ValueError('%d recorded steps exceed the episode bucket T = %d' % (len(actions), self.T))
            if any(a.shape != (len(episodes),) for a in actions):
                raise ValueError('actions must hold one index per episode and step')
        p = EpisodePlanner(self, episodes, imitation=actions is None)
        for t in range(self.T):
            p.build_step()
            p.advance(None if actions is None else (actions[t] if t < len(actions) else np.zeros(len(episodes), np.int64)))
        return p.finish()

    # ---- device -------------------------------------------------------------------------------------------------------------
    def _panoramas(self, model, t_, extras, k, n, B):
        """panorama encoder over the n panoramas of the tables with prefix k ('s<t>_': B of one step, 'all_': T * B)."""
        from collections import defaultdict
        from . import hipops
        fts = hipops.gather_segmean(self.features.dev, t_[k + 'feat_idx'], t_[k + 'feat_start'], None, n * self.W, None)
        pin = {'view_img_fts': fts.view(n, self.W, -1), 'loc_fts': t_[k + 'loc_fts'], 'nav_types': t_[k + 'nav_types'],
               'view_lens': t_[k + 'view_lens'], 'already_dropout': False}
        if self.objects is not None:
            ofts = hipops.gather_segmean(self.objects._fs.dev, t_[k + 'obj_idx'], t_[k + 'obj_start'], None, n * self.O, None)
            pin.update({'reverie_obj_img_fts': ofts.view(n, self.O, -1), 'reverie_obj_lens': t_[k + 'reverie_obj_lens'],
                        'reverie_obj_names': t_[k + 'reverie_obj_names'],
                        'reverie_obj_concat': (t_[k + 'ocat_idx'], t_[k + 'ocat_start'], t_[k + 'ocat_inv_idx'], t_[k + 'ocat_inv_start'])})
        for name, z in (extras or {}).get('panorama', {}).items():      # per-sample dictionary copies ([B, K, ...]) follow the joint batch
            pin[name] = z.repeat(n // B, *([1] * (z.dim() - 1))) if (torch.is_tensor(z) and n != B and z.dim() > 1 and z.shape[0] == B) else z
        return model('panorama', defaultdict(lambda: None, pin))

    @staticmethod
    def _nav_extras(extras, dtype):
        nav_extras = dict((extras or {}).get('navigation', {}))
        for name in ('front_vp_feats', 'front_gmap_feats', 'front_txt_feats'):      # constant over the episode: cast once, not per step
            if torch.is_tensor(nav_extras.get(name)) and nav_extras[name].is_floating_point():
                nav_extras[name] = nav_extras[name].to(dtype)
        return nav_extras

    def _nav_step(self, model, t_, s, txt, txt_kv, pano, pmask, fused, pool, last, nav_extras):
        """navigation step s over the tables 's<s>_*': node embeddings gathered from `pool` (the panoramas of this and all earlier steps,
        appended to here) and the previous [MEM] state `last`.  -> (logits of the configured fusion, the new [MEM] state, object logits | None)"""
        from collections import defaultdict
        from . import hipops
        k = 's%d_' % s
        B = pano.shape[0]
        if fused is None:
            fused = torch.sum(pano * pmask.unsqueeze(2), 1) / torch.sum(pmask, 1, keepdim=True)
        H = pano.shape[-1]
        pool += [pano.reshape(B * self.WT, H), fused.to(pano.dtype)]
        src = torch.cat(pool + ([last.to(pano.dtype)] if last is not None else []), 0)
        G = t_[k + 'gmap_step_ids'].shape[1]
        gimg = hipops.gather_segmean(src, t_[k + 'csr_idx'], t_[k + 'csr_start'], t_[k + 'csr_scale'], B * G,
                                     (t_[k + 'inv_idx'], t_[k + 'inv_start'], t_[k + 'inv_w'])).view(B, G, H)
        zero = pano.new_zeros(B, 1, H)
        memtok = zero if last is None else last.unsqueeze(1).to(pano.dtype)
        nin = {'txt_embeds': txt, 'txt_masks': t_['txt_masks'], 'gmap_img_embeds': gimg,
               'vp_img_embeds': torch.cat([zero, memtok, pano], 1), 'flops_count': False, 'txt_kv': txt_kv,
               'vp_obj_masks': t_[k + 'vp_obj_masks'] if self.objects is not None else None, 'nav_fusion': t_[k + 'nav_fusion']}
        for name in ('gmap_step_ids', 'gmap_pos_fts', 'gmap_pair_dists', 'gmap_visited_masks', 'gmap_masks', 'vp_pos_fts', 'vp_masks',
                     'vp_nav_masks'):
            nin[name] = t_[k + name]
        nin.update(nav_extras)
        out = model('navigation', defaultdict(lambda: None, nin))
        logits = {'local': out['local_logits'], 'global': out['global_logits']}.get(self.fusion, out['fused_logits'])
        # (the object logits are RETURNED, not kept on self: a tensor of a warm-up pass alive at capture time keeps that pass's autograd
        #  graph alive, whose AccumulateGrad nodes are bound to the warm-up stream — the captured backward then accumulates off-graph)
        return logits, out['cls_embeds'], out.get('obj_logits')

    def body(self, model, bufs, extras=None, hoist_text_kv=True, hoist_pano=True):
        """forward + imitation loss of the planned episodes from the tensors of `bufs` (EpisodeBuffers.t): no host data, no
        device -> host copy.  -> loss (sum over steps and samples of the cross-entropy / B, M/r2r/agent.py:664-667).
        hoist_pano: ONE panorama-encoder call over the T * B panoramas of the whole walk instead of one per step — the encoder sees
        the observation only (M/r2r/agent.py:548-556), and with teacher forcing every observation is known before the first step, so
        its T small launches-bound passes (B * W = 456 rows at B = 12) become one of T * B * W rows; the [MEM]-carrying navigation
        steps stay sequential.  Same embeddings and gradients (summation order aside; dropout draws differ)."""
        from collections import defaultdict
        from . import hipops
        dd = lambda d: defaultdict(lambda: None, d)
        extras = extras or {}
        t_ = bufs.t
        B = t_['txt_ids'].shape[0]
        lang = {'txt_ids': t_['txt_ids'], 'txt_masks': t_['txt_masks']}
        lang.update(extras.get('language', {}))
        pool, last, loss, ce_rows = [], None, 0.0, []
        panoramas = lambda k, n: self._panoramas(model, t_, extras, k, n, B)

        # (the hoisted panorama pass depends on the observations only: a parallel branch of the instruction encoder in the captured graph)
        with hipops.Branch('pano', 'nav_pano') as bp:
            whole = panoramas('all_', self.T * B) if hoist_pano else None
        txt = model('language', dd(lang))
        # (one autograd handle per step on the instruction states and their hoisted K|V projections: hipops.fanout)
        txt_h = hipops.fanout(txt, self.T + 1)
        txt_kv = model('text_kv', {'txt_embeds': txt_h[self.T]}) if hoist_text_kv else None
        kv_h = hipops.fanout_tree(txt_kv, self.T) if txt_kv is not None else [None] * self.T
        if whole is not None:
            bp.join(*whole)
            # (unbind: ONE backward node stacks the per-step gradients — not T zero-filled slice_backward tensors and T - 1 adds)
            whole_s = [None if x is None else x.view(self.T, B, *x.shape[1:]).unbind(0) for x in whole]
        nav_extras = self._nav_extras(extras, txt.dtype)
        for s in range(self.T):
            k = 's%d_' % s
            if whole is not None:
                pano, pmask, fused = (None if x is None else x[s] for x in whole_s)
            else:
                pano, pmask, fused = panoramas(k, B)
            logits, last, obj_logits = self._nav_step(model, t_, s, txt_h[s], kv_h[s], pano, pmask, fused, pool, last, nav_extras)
            if self.ignoreid < 0:
                ce_rows.append(hipops.cross_entropy_rows(logits, t_[k + 'target'], self.ignoreid))          # (summed once behind the loop)
                if self.objects is not None:        # object grounding at the goal viewpoints (M/reverie/agent_obj_goat.py:705-707)
                    ce_rows.append(hipops.cross_entropy_rows(obj_logits, t_[k + 'obj_target'], self.ignoreid))
            else:
                loss = loss + torch.nn.functional.cross_entropy(logits.float(), t_[k + 'target'], reduction='sum', ignore_index=self.ignoreid)
                if self.objects is not None:
                    loss = loss + torch.nn.functional.cross_entropy(obj_logits.float(), t_[k + 'obj_target'], reduction='sum', ignore_index=self.ignoreid)
        if ce_rows:
            loss = loss + torch.stack(ce_rows, 0).sum()
        return loss / B


def _obj_concat_tables(view_lens, obj_lens, V, O, W, prefix):
    """the [views | objects] row assembly of n panoramas (graphmap.build_obj_concat_index) and its inverse as fixed-size tables"""
    from . import graphmap
    n = len(view_lens)
    ci = graphmap.build_obj_concat_index(view_lens, obj_lens, V, O, W)
    inv = graphmap.inverse_index(ci[0], ci[1], None, n * V + n * O)
    n_tok = int(ci[1][-1])
    return {prefix + 'ocat_idx': _pad1np(ci[0].numpy()[:n_tok], n * W, -1, np.int32), prefix + 'ocat_start': ci[1],
            prefix + 'ocat_inv_idx': _pad1np(inv[0].numpy()[:n_tok], n * W, -1, np.int32), prefix + 'ocat_inv_start': inv[1]}


class EpisodePlanner:
    """TeacherEpisode.plan one step at a time: build_step() makes the host tables of step t from the navigator's current state,
    advance(actions) moves it (the teacher's actions when `imitation`, else the given ones — a sampled rollout decides them from the
    step's logits), finish() adds the joint panorama tables and returns the plan dict.  While a planner is alive it owns te.sim."""

    def __init__(self, te, episodes, imitation=True):
        self.te, self.imitation = te, imitation
        obs = te.sim.reset(episodes)
        self.B = B = len(obs)
        self.gmaps = [GraphMap(ob['viewpoint']) for ob in obs]
        for g, ob in zip(self.gmaps, obs):
            g.update_graph(ob)
        lang = language_inputs(obs)
        if lang['txt_ids'].shape[1] > te.L:
            raise ValueError('instruction of %d tokens exceeds the text bucket %d' % (lang['txt_ids'].shape[1], te.L))
        ids = torch.zeros(B, te.L, dtype=torch.int64)
        msk = torch.zeros(B, te.L, dtype=torch.bool)
        ids[:, :lang['txt_ids'].shape[1]], msk[:, :lang['txt_masks'].shape[1]] = lang['txt_ids'], lang['txt_masks']
        self.out = {'txt_ids': ids, 'txt_masks': msk}
        self.obs = obs
        self.store = NodeEmbedStore(B)
        self.ended = np.zeros(B, bool)
        self.traj = [{'instr_id': ob['instr_id'], 'path': [[ob['viewpoint']]]} for ob in obs]
        self.n_traj = 0
        self.t = 0
        self._gin = self._target = None

    def build_step(self):
        """-> {key: tensor} of step t (also kept for finish())."""
        from . import graphmap, nav_model
        te, t, obs, gmaps, ended, store, B = self.te, self.t, self.obs, self.gmaps, self.ended, self.store, self.B
        if t >= te.T:
            raise ValueError('the episode bucket holds %d steps' % te.T)
        afs = te.sim.angle_feat_size
        out = {}
        for i, g in enumerate(gmaps):
            if not ended[i]:
                g.node_step_ids[obs[i]['viewpoint']] = t + 1
        self.n_traj += int((~ended).sum())
        has_obj = te.objects is not None
        pano = panorama_inputs(obs, afs, te.W, te.O if has_obj else None)
        store.advance(B, te.WT)
        for i, g in enumerate(gmaps):
            if not ended[i]:
                store.rewrite(i, obs[i]['viewpoint'])
                for j, cvp in enumerate(pano['cand_vpids'][i]):
                    if not g.graph.visited(cvp):
                        store.accumulate(i, cvp, j)
        G = te.gw(t)
        gin = gmap_inputs(obs, gmaps, G, afs, mem_selectable=has_obj)
        vin = vp_inputs(obs, gmaps, pano['cand_vpids'], pano['view_lens'], pano['nav_types'], te.WT + 2, afs,
                        gmap_pos=(gin['gmap_vpids'], gin['gmap_pos_fts'].numpy()), obj_lens=pano['reverie_obj_lens'] if has_obj else None)
        # (the REVERIE agent has the shortest-path expert only, M/reverie/agent_obj_goat.py:390-417)
        target = teacher_action(obs, gin['gmap_vpids'], ended, gin['gmap_visited_masks'].numpy(), self.imitation and not has_obj, t, te.ignoreid)
        k = 's%d_' % t
        # feature gather of the panorama tokens: compact CSR (padding slots = empty segments)
        rows = pano['view_rows'].reshape(-1).numpy()
        valid = rows >= 0
        fidx = np.full(rows.shape[0], -1, np.int32)
        fidx[:int(valid.sum())] = rows[valid]
        out[k + 'feat_idx'] = torch.from_numpy(fidx)
        out[k + 'feat_start'] = torch.from_numpy(np.concatenate([[0], np.cumsum(valid)]).astype(np.int32))
        for name in ('loc_fts', 'nav_types', 'view_lens'):
            out[k + name] = pano[name]
        for name in ('gmap_step_ids', 'gmap_pos_fts', 'gmap_pair_dists', 'gmap_visited_masks', 'gmap_masks'):
            out[k + name] = gin[name]
        for name in ('vp_pos_fts', 'vp_masks', 'vp_nav_masks'):
            out[k + name] = vin[name]
        if has_obj:
            orow = pano['obj_rows'].reshape(-1).numpy()
            ov = orow >= 0
            oidx = np.full(orow.shape[0], -1, np.int32)
            oidx[:int(ov.sum())] = orow[ov]
            out[k + 'obj_idx'] = torch.from_numpy(oidx)
            out[k + 'obj_start'] = torch.from_numpy(np.concatenate([[0], np.cumsum(ov)]).astype(np.int32))
            out[k + 'reverie_obj_lens'], out[k + 'reverie_obj_names'] = pano['reverie_obj_lens'], pano['reverie_obj_names']
            out[k + 'vp_obj_masks'] = vin['vp_obj_masks']
            out[k + 'obj_target'] = torch.from_numpy(teacher_object(obs, ended, pano['view_lens'], te.ignoreid))
            out.update(_obj_concat_tables(pano['view_lens'], pano['reverie_obj_lens'], te.W, te.O, te.WT, k))
        out[k + 'nav_fusion'] = nav_model.nav_fusion_matrix(vin['vp_cand_vpids'], gin['gmap_vpids'], gin['gmap_visited_masks'], G, te.WT + 2)
        # host-side label validation, as train_step.collate_indices does for pre-training (goat_ce_fwd cannot raise; an
        # out-of-range target would only show up as a NaN loss): a teacher action is a map slot of this step or the ignore value
        if ((target >= G) | ((target < 0) & (target != te.ignoreid))).any():
            raise ValueError('teacher action outside the %d map slots of step %d: %s' % (G, t, target.tolist()))
        out[k + 'target'] = torch.from_numpy(target)
        # node embeddings: CSR over the pool of this and all earlier steps (+ the previous [MEM] state behind it)
        n_src = store.rows + (B if t > 0 else 0)
        mem_rows = [store.rows + b for b in range(B)] if t > 0 else None
        idx, start, scale = store.csr(gin['gmap_vpids'], G, mem_rows)
        inv = graphmap.inverse_index(idx, start, scale, n_src)
        n_tok = int(start[-1])
        out[k + 'csr_idx'] = _pad1np(idx[:n_tok] if n_tok else idx[:0], n_src, -1, np.int32)
        out[k + 'csr_start'], out[k + 'csr_scale'] = torch.from_numpy(start), torch.from_numpy(scale)
        out[k + 'inv_idx'] = _pad1np(inv[0].numpy()[:n_tok], n_src, -1, np.int32)
        out[k + 'inv_start'] = inv[1]
        out[k + 'inv_w'] = _pad1np(inv[2].numpy()[:n_tok], n_src, 0.0, np.float32)
        self._gin, self._target = gin, target
        self.out.update(out)
        return out

    def advance(self, actions=None):
        """the move of step t: the teacher's (imitation) or `actions` [B] (indices into the step's map nodes, 0 = [stop])."""
        te, t, obs, gmaps, ended, B = self.te, self.t, self.obs, self.gmaps, self.ended, self.B
        gin, target = self._gin, self._target
        if gin is None:
            raise RuntimeError('advance() before build_step()')
        if not self.imitation and actions is None:
            raise ValueError('a planner of a sampled walk needs the actions of every step')
        moves = []
        for i in range(B):
            stop = obs[i]['viewpoint'] == obs[i]['gt_path'][-1]
            if stop or ended[i] or gin['no_vp_left'][i] or t == te.T - 1:
                moves.append(None)
                continue
            if self.imitation:
                a = int(target[i])
            else:
                a = int(actions[i])
                if not 0 <= a < len(gin['gmap_vpids'][i]) or (a > 0 and bool(gin['gmap_visited_masks'][i, a])):
                    raise ValueError('step %d, episode %d: recorded action %d is not a navigable node of the map' % (t, i, a))
            nxt = gin['gmap_vpids'][i][a]
            if nxt is None:             # a recorded [stop] (node 0): the episode ends here (M/r2r/agent.py:661)
                moves.append(None)
            else:
                hop = gmaps[i].graph.path(obs[i]['viewpoint'], nxt)
                self.traj[i]['path'].append(hop)
                prev = self.traj[i]['path'][-2][-1] if len(hop) == 1 else hop[-2]
                view = next(c['pointId'] for c in obs[i]['scan_graph'].candidates(prev) if c['viewpointId'] == nxt)
                moves.append((nxt, view))
        self.obs = obs = te.sim.step(moves)
        for i, ob in enumerate(obs):
            if not ended[i]:
                gmaps[i].update_graph(ob)
        self.ended = np.logical_or(ended, np.array([m is None for m in moves]))
        self._gin = self._target = None
        self.t = t + 1

    def finish(self):
        """the plan of the whole episode (every step built and advanced)."""
        te, out, B = self.te, self.out, self.B
        if self.t != te.T:
            raise RuntimeError('finish() after %d of %d steps' % (self.t, te.T))
        # the panoramas of all steps as ONE batch [T * B] (body(hoist_pano=True)): the compact feature-gather CSR of the steps joined
        valid_rows, starts, off = [], [], 0
        for t in range(te.T):
            st = out['s%d_feat_start' % t].numpy()
            valid_rows.append(out['s%d_feat_idx' % t].numpy()[:int(st[-1])])
            starts.append(st[:-1] + off)
            off += int(st[-1])
        out['all_feat_idx'] = _pad1np(np.concatenate(valid_rows), te.T * B * te.W, -1, np.int32)
        out['all_feat_start'] = torch.from_numpy(np.concatenate(starts + [[off]]).astype(np.int32))
        for name in ('loc_fts', 'nav_types', 'view_lens'):
            out['all_' + name] = torch.cat([out['s%d_%s' % (t, name)] for t in range(te.T)], 0)
        if te.objects is not None:
            valid_rows, starts, off = [], [], 0
            for t in range(te.T):
                st = out['s%d_obj_start' % t].numpy()
                valid_rows.append(out['s%d_obj_idx' % t].numpy()[:int(st[-1])])
                starts.append(st[:-1] + off)
                off += int(st[-1])
            out['all_obj_idx'] = _pad1np(np.concatenate(valid_rows), te.T * B * te.O, -1, np.int32)
            out['all_obj_start'] = torch.from_numpy(np.concatenate(starts + [[off]]).astype(np.int32))
            for name in ('reverie_obj_lens', 'reverie_obj_names'):
                out['all_' + name] = torch.cat([out['s%d_%s' % (t, name)] for t in range(te.T)], 0)
            out.update(_obj_concat_tables(out['all_view_lens'], out['all_reverie_obj_lens'], te.W, te.O, te.WT, 'all_'))
        out['_traj'], out['_n_traj'] = self.traj, self.n_traj
        return out


class _RowIndex:
    """key -> row lookup of a feature store without its table (what GraphSim needs of it on the host)."""

    def __init__(self, keys):
        self.index = {k: i for i, k in enumerate(keys)}

    def row(self, scan, vp):
        return self.index['%s_%s' % (scan, vp)]


def _plan_worker_main(conn, spec):
    import os
    import time
    os.environ.setdefault('HIP_VISIBLE_DEVICES', '-1')        # host work only: the worker never touches the GPU
    torch.set_num_threads(1)          # small host tensors only: the default intra-op pool (one thread per core: 128-256 on the GPU boxes) costs
                                      # tens of ms in wake-ups per plan (measured: 60 ms per plan with the pool, 13 ms without)
    scans = spec['scans']
    te = TeacherEpisode(GraphSim(_RowIndex(spec['keys']), spec['angle_feat_size'], objects=spec.get('objects')), None, spec['n_steps'],
                        spec['text_len'], pano_width=spec['pano_width'], gmap_width=lambda t, w=spec['gmap_widths']: w[min(t, len(w) - 1)],
                        fusion=spec['fusion'], ignoreid=spec['ignoreid'], obj_width=spec.get('obj_width', 20))
    while True:
        try:
            episodes = conn.recv()
        except EOFError:
            break
        if episodes is None:
            break
        try:
            t0 = time.perf_counter()
            plan = te.plan([dict(e, scan=scans[e['scan']]) for e in episodes])
            plan['_plan_s'] = time.perf_counter() - t0
            plan.pop('_traj', None)
            # numpy through the pipe: torch tensors would travel as one shared-memory segment + file descriptor EACH (~150 per plan: 10 ms)
            conn.send({k: (('__t', v.numpy()) if torch.is_tensor(v) else v) for k, v in plan.items()})
        except Exception as e:      # noqa: BLE001  (reported to the caller, the worker stays alive)
            conn.send(e)


class PlanWorker:
    """TeacherEpisode.plan in a worker PROCESS.  The plan of a batch of episodes is 10-15 ms of pure-Python table building; next to it the
    training process copies the previous plan into the pinned buffer and launches a ~3 000-node episode graph (several ms of host time in
    hipGraphLaunch).  A thread does not help (the builders hold the GIL: 21-24 ms per episode against 19 ms of GPU work); a process does.
    submit(episodes) returns at once (the episodes travel with their scan NAMES; the worker holds the ScanGraphs, its own GraphSim over the
    store's key -> row map, and never initialises the GPU — spawn context, safe beside an initialised HIP runtime); a reader thread of this
    process drains the worker's pipe (plans are ~3 MB: a worker blocked in send while this process blocks in submit would deadlock);
    result() hands back the next plan dict (CPU tensors) in submission order.

        pw = PlanWorker(te, store.keys, scans);  pw.submit(eps_0); pw.submit(eps_1)          # two in flight
        for k in ...:  plan = pw.result(); pw.submit(eps_k2); bufs.load(plan); graph.replay()"""

    def __init__(self, te, keys, scans):
        import multiprocessing as mp
        import queue
        import threading
        ctx = mp.get_context('spawn')
        self.conn, child = ctx.Pipe()
        scans = {sc.name: sc for sc in (scans.values() if isinstance(scans, dict) else scans)}
        spec = {'keys': list(keys), 'scans': scans, 'angle_feat_size': te.sim.angle_feat_size, 'n_steps': te.T, 'text_len': te.L,
                'pano_width': te.W, 'gmap_widths': [int(te.gw(t)) for t in range(max(te.T, 1))], 'fusion': te.fusion, 'ignoreid': te.ignoreid,
                'objects': te.objects.meta() if te.objects is not None else None, 'obj_width': te.O}      # (REVERIE: object metadata, no features)
        self.proc = ctx.Process(target=_plan_worker_main, args=(child, spec), daemon=True)
        self.proc.start()
        child.close()
        self.pending = 0
        self._q = queue.Queue()

        def drain():
            while True:
                try:
                    self._q.put(self.conn.recv())
                except (EOFError, OSError):
                    self._q.put(EOFError('plan worker exited'))
                    return
        self._reader = threading.Thread(target=drain, daemon=True)
        self._reader.start()

    def submit(self, episodes):
        self.conn.send([dict(e, scan=e['scan'].name if not isinstance(e['scan'], str) else e['scan']) for e in episodes])
        self.pending += 1

    def result(self, timeout=120.0):
        import queue
        if self.pending <= 0:
            raise RuntimeError('PlanWorker.result() without a submitted batch of episodes')
        try:
            plan = self._q.get(timeout=timeout)
        except queue.Empty:
            raise TimeoutError('plan worker (pid %s, %s) returned no plan within %.0f s'
                               % (self.proc.pid, 'alive' if self.proc.is_alive() else 'exited', timeout)) from None
        self.pending -= 1
        if isinstance(plan, Exception):
            raise plan
        return {k: (torch.from_numpy(v[1]) if (isinstance(v, tuple) and len(v) == 2 and isinstance(v[0], str) and v[0] == '__t') else v)
                for k, v in plan.items()}

    def close(self):
        """stop the worker (idempotent); plans not yet fetched are dropped."""
        if self.proc is None:
            return
        try:
            self.conn.send(None)
        except (OSError, BrokenPipeError):
            pass
        self.proc.join(5)
        if self.proc.is_alive():
            self.proc.terminate()
            self.proc.join(5)
        self.conn.close()
        self.proc = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def _pad1np(a, n, fill, dtype):
    out = np.full(n, fill, dtype)
    out[:len(a)] = a
    return torch.from_numpy(out)


class EpisodeBuffers:
    """the tensors of a TeacherEpisode.plan at FIXED device addresses (one flat buffer; one pinned H2D copy per new plan)."""
    ALIGN = 256

    def __init__(self, plan, device='cuda'):
        self.device = torch.device(device)
        self.layout, off = [], 0
        for k in sorted(plan):
            v = plan[k]
            if torch.is_tensor(v):
                self.layout.append((k, off, tuple(v.shape), v.dtype))
                off += (v.numel() * v.element_size() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.nbytes = max(off, self.ALIGN)
        self.flat = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self.host = torch.zeros(self.nbytes, dtype=torch.uint8)
        if self.device.type == 'cuda':
            self.host = self.host.pin_memory()
        self.t = {}
        for k, off, shape, dtype in self.layout:
            n = 1
            for s in shape:
                n *= s
            self.t[k] = self.flat[off:off + n * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(shape)
        self._copied = None
        self.load(plan)

    def load_part(self, part):
        """the tensors of `part` only (e.g. EpisodePlanner.build_step(): the tables of one step): packed into the pinned buffer, ONE H2D of
        the byte range that covers them (the bytes between them are what the device already holds), their memoised masks refreshed."""
        from . import layers
        if self._copied is not None:
            self._copied.synchronize()
        where = getattr(self, '_where', None)
        if where is None:
            where = self._where = {k: (off, shape, dtype) for k, off, shape, dtype in self.layout}
        dst = self.host.numpy()
        lo, hi = self.nbytes, 0
        for k, v in part.items():
            if not torch.is_tensor(v):
                continue
            off, shape, dtype = where[k]
            if tuple(v.shape) != shape or v.dtype != dtype:
                raise ValueError('EpisodeBuffers: %s is %s %s, the captured layout has %s %s' % (k, tuple(v.shape), v.dtype, shape, dtype))
            n = v.numel() * v.element_size()
            if n:
                dst[off:off + n] = v.contiguous().view(-1).view(torch.uint8).numpy()
                lo, hi = min(lo, off), max(hi, off + n)
        if hi > lo:
            self.flat[lo:hi].copy_(self.host[lo:hi], non_blocking=True)
            if self.device.type == 'cuda':
                self._copied = torch.cuda.Event()
                self._copied.record()
        layers.refresh_masks(only=[self.t[k] for k in part if k in self.t])

    def load(self, plan, stream=None):
        """pack `plan` into the pinned buffer and copy it to the device (asynchronously on `stream` / the current stream)."""
        from . import layers
        if self._copied is not None:
            self._copied.synchronize()              # the previous H2D has read the pinned buffer
        dst = self.host.numpy()
        for k, off, shape, dtype in self.layout:
            v = plan.get(k)
            if v is None or tuple(v.shape) != shape or v.dtype != dtype:
                raise ValueError('EpisodeBuffers: %s is %s %s, the captured layout has %s %s'
                                 % (k, None if v is None else tuple(v.shape), None if v is None else v.dtype, shape, dtype))
            n = v.numel() * v.element_size()
            if n:
                dst[off:off + n] = v.contiguous().view(-1).view(torch.uint8).numpy()
        if stream is not None:
            with torch.cuda.stream(stream):
                self.flat.copy_(self.host, non_blocking=True)
        else:
            self.flat.copy_(self.host, non_blocking=True)
        if self.device.type == 'cuda':
            self._copied = torch.cuda.Event()
            self._copied.record(stream)
        layers.refresh_masks()



class SampledEpisode:
    """Pass 1 of the two-pass sampled rollout (feedback = 'sample': M/r2r/agent.py:436-437,575-607) as captured FORWARD graphs over
    the buffers of a TeacherEpisode: one graph for the instruction and its K|V projections, one per step t (feature gather, panorama
    encoder, node-embedding gather over the panoramas of the steps <= t, navigation step, softmax).  Nothing is differentiated here:
    the pass only decides the walk.  Per step the host builds the step's tables from the navigator's state (EpisodePlanner.build_step),
    copies them into the episode buffers (one small pinned H2D), replays the step graph, reads the B x G action probabilities back — the
    one device -> host copy of the step — and samples.  run() returns the finished plan (== TeacherEpisode.plan(episodes, actions):
    the walk the policy took, the DAgger labels of its states), ready for `bufs.load(plan)` + the captured forward + backward body.

        se = SampledEpisode(te, model, bufs, extras)              # captures T + 1 small graphs (bufs holds any valid plan)
        plan, actions = se.run(episodes, rng);  bufs.load(plan);  episode_graph.replay()"""

    def __init__(self, te, model, bufs, extras=None, bump_masks=True):
        """bump_masks=False: the step graphs do not advance the device-side dropout counter (they read whatever it holds): for running
        this pass BESIDE a training graph on another stream — a bump between that graph's forward and backward kernels would make its
        backward regenerate other masks than its forward drew; a forward-only pass needs no particular counter value."""
        from collections import defaultdict
        from . import hipops
        self.te, self.bufs = te, bufs
        extras = extras or {}
        t_ = bufs.t
        B = t_['txt_ids'].shape[0]
        dd = lambda d: defaultdict(lambda: None, d)

        def fresh_masks():                  # (a replay draws new dropout masks: the in-graph bump of the device-side counter, if one is in use)
            if bump_masks and hipops.RngState.dev is not None:
                hipops.RngState.dev.add_(0x9E3779B1)

        def language():
            fresh_masks()
            lang = {'txt_ids': t_['txt_ids'], 'txt_masks': t_['txt_masks']}
            lang.update(extras.get('language', {}))
            txt = model('language', dd(lang))
            return txt, model('text_kv', {'txt_embeds': txt})
        self.g_lang, (self.txt, self.txt_kv) = self._capture(language)
        nav_extras = te._nav_extras(extras, self.txt.dtype)
        self.g_step, self.probs = [], []
        pool, last = [], None
        for s in range(te.T):
            def step(s=s, pool=pool, last=last):
                fresh_masks()
                mine = list(pool)
                pano, pmask, fused = te._panoramas(model, t_, extras, 's%d_' % s, B, B)
                logits, new_last, _ = te._nav_step(model, t_, s, self.txt, self.txt_kv, pano, pmask, fused, mine, last, nav_extras)
                return torch.softmax(logits.float(), 1), mine, new_last
            g, (probs, pool, last) = self._capture(step)
            self.g_step.append(g)
            self.probs.append(probs)
        self._keep = (pool, last)               # (the static outputs the later step graphs read)
        self.host_s = 0.0

    @staticmethod
    def _capture(fn):
        from . import hipops
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), hipops.graph(g):
            out = fn()
        return g, out

    @staticmethod
    def sample(probs, rng):
        """Categorical(probs).sample() on the host: inverse CDF per row (rows are the B episodes; never a zero-probability node)."""
        c = np.cumsum(probs.astype(np.float64), 1)
        # u < c[-1] strictly (a product that rounds up to c[-1] would select the slot behind the last node with probability mass —
        # a padding slot, which advance() rejects): the first index whose cumulative mass exceeds u always has probs > 0
        u = np.minimum(rng.random_sample(probs.shape[0]) * c[:, -1], np.nextafter(c[:, -1], 0.0))
        return np.array([int(np.searchsorted(c[i], u[i], side='right')) for i in range(probs.shape[0])], np.int64)

    def run(self, episodes, rng=None, sampler=None):
        """-> (plan, actions).  sampler(t, probs [B, G] numpy) -> actions [B] overrides the random draw (tests: a fixed action sequence)."""
        import time
        te, bufs = self.te, self.bufs
        rng = rng if rng is not None else np.random.RandomState(0)
        t0 = time.perf_counter()
        p = EpisodePlanner(te, episodes, imitation=False)
        self.host_s = time.perf_counter() - t0
        bufs.load_part({'txt_ids': p.out['txt_ids'], 'txt_masks': p.out['txt_masks']})
        self.g_lang.replay()
        actions = []
        zeros = np.zeros(len(episodes), np.int64)
        for t in range(te.T):
            t0 = time.perf_counter()
            part = p.build_step()
            self.host_s += time.perf_counter() - t0
            bufs.load_part(part)
            self.g_step[t].replay()
            probs = self.probs[t].cpu().numpy()             # (synchronises: the step's one read-back)
            a = np.asarray(sampler(t, probs), np.int64) if sampler is not None else self.sample(probs, rng)
            a = np.where(p.ended, 0, a)
            actions.append(a)
            t0 = time.perf_counter()
            p.advance(a)
            self.host_s += time.perf_counter() - t0
            if p.ended.all():
                break
        self.steps = len(actions)
        t0 = time.perf_counter()
        while p.t < te.T:                                     # every episode has ended: the remaining steps carry no label
            p.build_step()
            p.advance(zeros)
        plan = p.finish()
        self.host_s += time.perf_counter() - t0
        return plan, actions


class SinglePassSampledEpisode:
    """The sampled half of a dagger iteration in ONE pass, as the reference runs it (M/r2r/agent.py:596-690: the action is sampled at
    :629-633 from the SAME forward whose logits carry the loss), at graph speed: the instruction graph and the T step graphs of
    SampledEpisode captured WITH their autograd state kept alive (activations stay in the graphs' shared memory pool), and ONE captured
    backward graph that forms the loss over the logits the steps left behind — the DAgger labels of the visited states are part of each
    step's tables — and differentiates it through all T steps ([MEM] state, node-embedding pool, the instruction's K|V bank).  Against the
    two-pass form (SampledEpisode + the episode graph) the forward of pass 2 is gone, and the walk and the gradient see the same dropout
    masks (sample-equivalent to the eager single pass, not only distribution-equivalent).

        sp = SinglePassSampledEpisode(te, model, bufs, extras, prologue=lambda: arena.zero('nav'))
        traj, actions = sp.run(episodes, rng)          # gradients of the sampled loss are in the arena / .grad; sp.loss holds its value

    Dropout: the device-side counter is bumped ONCE per iteration (instruction graph); every op of every step draws from its own counter
    range, and the backward graph regenerates the masks from the same counter value.  Every step graph is replayed in every iteration (an
    ended episode's rows carry the ignore label): no graph ever differentiates through buffers it has not written.
    As for every capture of a training step: no tensor of an EARLIER backward pass (a loss, logits kept on an object) may be alive when
    this object is built — it would keep AccumulateGrad nodes bound to the eager stream alive and the captured backward would accumulate
    outside the capture (hipops.graph raises; on ROCm 7.2 ending that capture can also crash the process)."""

    def __init__(self, te, model, bufs, extras=None, prologue=None, loss_scale=1.0):
        import gc
        from collections import defaultdict
        from . import hipops
        self.te, self.bufs = te, bufs
        extras = extras or {}
        t_ = bufs.t
        B, T = t_['txt_ids'].shape[0], te.T
        dd = lambda d: defaultdict(lambda: None, d)

        def language():
            if prologue is not None:
                prologue()
            if hipops.RngState.dev is not None:
                hipops.RngState.dev.add_(0x9E3779B1)
            lang = {'txt_ids': t_['txt_ids'], 'txt_masks': t_['txt_masks']}
            lang.update(extras.get('language', {}))
            txt = model('language', dd(lang))
            txt_h = hipops.fanout(txt, T + 1)            # one autograd handle per step on the instruction states and their K|V projections
            kv_h = hipops.fanout_tree(model('text_kv', {'txt_embeds': txt_h[T]}), T)
            return txt_h, kv_h

        def step(s, txt_h, kv_h, pool, last, nav_extras):
            mine = list(pool)
            pano, pmask, fused = te._panoramas(model, t_, extras, 's%d_' % s, B, B)
            logits, new_last, obj_logits = te._nav_step(model, t_, s, txt_h[s], kv_h[s], pano, pmask, fused, mine, last, nav_extras)
            with torch.no_grad():
                probs = torch.softmax(logits.detach().float(), 1)
            return probs, mine, new_last, logits, obj_logits

        def backward(logits, obj_logits):
            rows = []
            for s in range(T):
                k = 's%d_' % s
                rows.append(hipops.cross_entropy_rows(logits[s], t_[k + 'target'], te.ignoreid))
                if te.objects is not None:          # object grounding at the goal viewpoints (M/reverie/agent_obj_goat.py:705-707)
                    rows.append(hipops.cross_entropy_rows(obj_logits[s], t_[k + 'obj_target'], te.ignoreid))
            loss = torch.stack(rows, 0).sum() / B
            (loss * loss_scale if loss_scale != 1.0 else loss).backward()
            return loss.detach()

        def whole():
            txt_h, kv_h = language()
            nav_extras = te._nav_extras(extras, txt_h[0].dtype)
            pool, last, lg, og = [], None, [], []
            for s in range(T):
                _, pool, last, l, o = step(s, txt_h, kv_h, pool, last, nav_extras)
                lg.append(l)
                og.append(o)
            return backward(lg, og)

        # eager warm-up of the whole chain (weight shadows, kernel attributes, GEMM tuning where the tuner is on) with the parallel branches
        # forked as the captures will fork them; nothing of it may outlive into the captures (hipops.graph raises on a live warm-up graph)
        # A tensor of an EARLIER backward pass that is still alive shows up right here, as torch's AccumulateGrad stream-mismatch warning (the
        # warm-up runs on a stream of its own): raised as an error BEFORE anything is captured — inside the capture the same condition ends in
        # a crash of hipStreamEndCapture on ROCm 7.2, not in an exception.
        import warnings
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)                 # (torch emits this warning ONCE per process otherwise: an earlier, harmless occurrence would hide this one)
        with warnings.catch_warnings():
            warnings.filterwarnings('error', message=".*AccumulateGrad node's stream does not match.*")
            try:
                with torch.cuda.stream(side), hipops.Branch.like_capture():
                    whole()
            except UserWarning as e:
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                hipops.WgradQueue.reset()
                raise RuntimeError('SinglePassSampledEpisode: an autograd graph of an earlier pass is still alive (a loss / logits tensor kept '
                                   'somewhere): its AccumulateGrad nodes are bound to that pass\'s stream and a captured backward would accumulate '
                                   'outside the capture.  Release those tensors (del, gc.collect()) before building this object.') from e
            finally:
                torch.set_warn_always(always)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gc.collect()

        self._pool = None
        self.g_lang, (txt_h, kv_h) = self._capture(language)
        nav_extras = te._nav_extras(extras, txt_h[0].dtype)
        self.g_step, self.probs = [], []
        pool, last, lg, og = [], None, [], []
        for s in range(T):
            g, (probs, pool, last, l, o) = self._capture(lambda s=s, pool=pool, last=last: step(s, txt_h, kv_h, pool, last, nav_extras))
            self.g_step.append(g)
            self.probs.append(probs)
            lg.append(l)
            og.append(o)
        self.g_bwd, self.loss = self._capture(lambda: backward(lg, og))
        self._keep = (txt_h, kv_h, pool, last, lg, og)         # (the static tensors the graphs read)
        torch.cuda.synchronize()
        self.host_s, self.steps = 0.0, 0

    def _capture(self, fn):
        from . import hipops
        g = torch.cuda.CUDAGraph()
        with hipops.graph(g, **({} if self._pool is None else {'pool': self._pool})):
            out = fn()
        self._pool = g.pool()
        return g, out

    def run(self, episodes, rng=None, sampler=None):
        """-> (trajectories, actions).  The parameter gradients of the sampled loss are where the model's backward puts them (the gradient
        arena / .grad) when this returns (asynchronously: on the current stream); `self.loss` is the loss of the iteration (device scalar).
        sampler(t, probs [B, G] numpy) -> actions [B] overrides the random draw (tests: a fixed action sequence)."""
        import time
        te, bufs = self.te, self.bufs
        rng = rng if rng is not None else np.random.RandomState(0)
        t0 = time.perf_counter()
        p = EpisodePlanner(te, episodes, imitation=False)
        self.host_s = time.perf_counter() - t0
        bufs.load_part({'txt_ids': p.out['txt_ids'], 'txt_masks': p.out['txt_masks']})
        self.g_lang.replay()
        actions, zeros = [], np.zeros(len(episodes), np.int64)
        self.steps = 0
        for t in range(te.T):
            t0 = time.perf_counter()
            part = p.build_step()
            self.host_s += time.perf_counter() - t0
            bufs.load_part(part)
            self.g_step[t].replay()
            if p.ended.all():                                 # nothing left to decide: the step runs on ignore labels, no read-back
                a = zeros
            else:
                probs = self.probs[t].cpu().numpy()           # (synchronises: the step's one read-back)
                a = np.asarray(sampler(t, probs), np.int64) if sampler is not None else SampledEpisode.sample(probs, rng)
                a = np.where(p.ended, 0, a)
                actions.append(a)
                self.steps += 1
            t0 = time.perf_counter()
            p.advance(a)
            self.host_s += time.perf_counter() - t0
        self.g_bwd.replay()
        self.n_traj = p.n_traj
        return p.traj, actions
