"""Host-side index building for the graph-map token assembly.

The reference assembles global-map tokens, local [stop]-prefixed tokens and the SAP logit fusion with
Python loops over viewpoint-id *strings* that launch thousands of tiny index/stack kernels
(GlobalMapEncoder._aggregate_gmap_features P/model/vilmodel_goat.py:430-468, vp_input_embedding :377-391,
forward_sap fusion loop P/model/pretrain_goat.py:329-345).  Here the strings are resolved ONCE per batch
on the host into small int32 index tensors; the device work is then one gather/segment-mean kernel
(goat_gather_segmean_*) and one tiny matmul.
"""
import numpy as np
import torch


def _to_list(x):
    if torch.is_tensor(x):
        return x.detach().cpu().tolist()
    return list(x)


def build_gmap_index(traj_step_lens, traj_vp_view_lens, traj_vpids, traj_cand_vpids, gmap_vpids, n_gmap, V, fused):
    """CSR description of gmap token sources.

    Source rows: [0, N*V) = panorama token (pano n, view j) at n*V+j ; [N*V, N*V+N) = fused panorama row n.
    Output token (b, g) -> segment b*n_gmap+g.  Semantics follow _aggregate_gmap_features exactly: a visited
    node takes the fused embedding of its (last) visit step (or the masked mean over views when fusion is
    off); an unvisited node the mean of every candidate-view embedding that saw it, where "visited" is
    evaluated progressively step by step as in the reference loop; g = 0 is the zero [stop] token.
    """
    step_lens = _to_list(traj_step_lens)
    view_lens = _to_list(traj_vp_view_lens)
    B = len(step_lens)
    N = sum(step_lens)
    idx, start, scale = [], [0], []
    n0 = 0
    for b in range(B):
        visited, unvisited = {}, {}
        for t in range(step_lens[b]):
            n = n0 + t
            visited[traj_vpids[b][t]] = n
            for j, vp in enumerate(traj_cand_vpids[b][t]):
                if vp not in visited:
                    unvisited.setdefault(vp, []).append(n * V + j)
        n0 += step_lens[b]
        for g in range(n_gmap):
            vps = gmap_vpids[b]
            if g == 0 or g >= len(vps):
                scale.append(1.0)
            else:
                vp = vps[g]
                if vp in visited:
                    n = visited[vp]
                    if fused:
                        idx.append(N * V + n)
                        scale.append(1.0)
                    else:
                        idx.extend(n * V + j for j in range(view_lens[n]))
                        scale.append(1.0 / view_lens[n])
                else:
                    rows = unvisited[vp]
                    idx.extend(rows)
                    scale.append(1.0 / len(rows))
            start.append(len(idx))
    if not idx:
        idx = [-1]
    return (torch.tensor(idx, dtype=torch.int32), torch.tensor(start, dtype=torch.int32),
            torch.tensor(scale, dtype=torch.float32))


def build_vp_index(traj_step_lens, traj_vp_view_lens, V):
    """Local-branch tokens: (b,0) = zero [stop]; (b,j>=1) = view j-1 of sample b's LAST panorama, for
    j < max_b(view_len_last)+1 — padded view slots included, as in the reference (pad_tensors_wgrad of
    x[-1] then [:max_vp_len], P/model/vilmodel_goat.py:378-388).  Pure arithmetic: numpy, no per-token Python."""
    step_lens = np.asarray(_to_list(traj_step_lens), dtype=np.int64)
    view_lens = np.asarray(_to_list(traj_vp_view_lens), dtype=np.int64)
    B = len(step_lens)
    last = np.cumsum(step_lens) - 1
    vp_lens = view_lens[last] + 1
    width = int(vp_lens.max())
    idx = (last[:, None] * V + np.arange(width - 1, dtype=np.int64)[None, :]).reshape(-1)
    per_tok = np.ones((B, width), dtype=np.int64)
    per_tok[:, 0] = 0                                  # the [stop] slot is an empty segment
    start = np.concatenate([[0], np.cumsum(per_tok.reshape(-1))])
    return (torch.from_numpy(idx.astype(np.int32)), torch.from_numpy(start.astype(np.int32)),
            torch.from_numpy(vp_lens.astype(np.int64)), width)


def build_sap_fusion(traj_cand_vpids, gmap_vpids, gmap_visited_masks, n_gmap, n_local):
    """0/1 matrix M [B, n_gmap, n_local]: fused[b,g] = global[b,g] + sum_j M[b,g,j] * local[b,j]
    (P/model/pretrain_goat.py:329-345)."""
    vis = _to_list(gmap_visited_masks)
    B = len(gmap_vpids)
    M = np.zeros((B, n_gmap, n_local), dtype=np.float32)
    for b in range(B):
        M[b, 0, 0] = 1.0
        visited = set(vp for vp, m in zip(gmap_vpids[b], vis[b]) if m)
        tmp, bw = {}, []
        for j, c in enumerate(traj_cand_vpids[b][-1]):
            if c in visited:
                bw.append(j + 1)
            else:
                tmp[c] = j + 1
        for g, vp in enumerate(gmap_vpids[b]):
            if g > 0 and vp not in visited:
                if vp in tmp:
                    M[b, g, tmp[vp]] += 1.0
                else:
                    for j in bw:
                        M[b, g, j] += 1.0
    return torch.from_numpy(M)


def build_obj_concat_index(view_lens, obj_lens, V, O, W):
    """REVERIE/SOON panorama rows: token j of row n is view j (j < view_len), object j - view_len
    (view_len <= j < view_len + obj_len) or padding — the per-row torch.cat + pad_tensors_wgrad of
    P/model/vilmodel_goat.py:331-340 as one gather.  Source rows: [0, N*V) views, [N*V, N*V + N*O) objects.
    -> (idx int32 [n_tokens], start int32 [N*W + 1]); padding slots are empty segments (zeros)."""
    vl = np.asarray(_to_list(view_lens), dtype=np.int64)
    ol = np.asarray(_to_list(obj_lens), dtype=np.int64)
    N = len(vl)
    over = np.nonzero(vl + ol > W)[0]
    if over.size:
        n = int(over[0])
        raise ValueError('row %d: %d views + %d objects exceed the padded width %d' % (n, vl[n], ol[n], W))
    J = np.arange(W, dtype=np.int64)[None, :]
    rows = np.arange(N, dtype=np.int64)[:, None]
    is_view = J < vl[:, None]
    is_obj = (J >= vl[:, None]) & (J < (vl + ol)[:, None])
    src = np.where(is_view, rows * V + J, N * V + rows * O + (J - vl[:, None]))
    used = is_view | is_obj
    idx = src[used]
    start = np.concatenate([[0], np.cumsum(used.reshape(-1))])
    if idx.size == 0:
        idx = np.array([-1])
    return torch.from_numpy(idx.astype(np.int32)), torch.from_numpy(start.astype(np.int32))


def inverse_index(idx, start, scale, n_src):
    """Inverse of a gather index (idx int32 [n_tok], start int32 [n_seg + 1], scale float32 [n_seg] | None): for every source row
    the segments that read it -> (inv_idx int32 [n_used] = segment ids grouped by source row, inv_start int32 [n_src + 1],
    inv_w float32 [n_used] = the segments' scales, or None when scale is None).  The gather's backward pass is then itself a
    gather over the output gradient (hipops.gather_segmean(..., inverse=...)): one writer per source row, no atomics."""
    idx = np.asarray(idx.cpu() if torch.is_tensor(idx) else idx, dtype=np.int64)
    start = np.asarray(start.cpu() if torch.is_tensor(start) else start, dtype=np.int64)
    n_seg = len(start) - 1
    seg_of_tok = np.repeat(np.arange(n_seg, dtype=np.int64), np.diff(start)) if n_seg else np.zeros(0, dtype=np.int64)
    idx = idx[:len(seg_of_tok)]                      # (an empty index is stored as [-1] with no segment pointing at it)
    used = idx >= 0
    src, seg = idx[used], seg_of_tok[used]
    if src.size and (src.max() >= n_src):
        raise ValueError('gather index refers to row %d of %d' % (int(src.max()), n_src))
    order = np.argsort(src, kind='stable')
    inv_idx = seg[order]
    counts = np.bincount(src, minlength=n_src) if src.size else np.zeros(n_src, dtype=np.int64)
    inv_start = np.concatenate([[0], np.cumsum(counts)])
    inv_w = None
    if scale is not None:
        sc = np.asarray(scale.cpu() if torch.is_tensor(scale) else scale, dtype=np.float32)
        inv_w = torch.from_numpy(sc[inv_idx].astype(np.float32)) if inv_idx.size else torch.zeros(1, dtype=torch.float32)
    if inv_idx.size == 0:
        inv_idx = np.array([-1])
    return torch.from_numpy(inv_idx.astype(np.int32)), torch.from_numpy(inv_start.astype(np.int32)), inv_w
