"""Oracle: scene-level pipeline around the model (test infrastructure only).

Restates reference ``dataset.py:56-67`` (tile grid), ``inferencer.py:52-110``
(tile batcher + mask fusion), ``graph_extraction.py:24-28,130-139`` +
``graph_utils.py:572-591`` (mask -> points, greedy radius NMS) and
``inferencer.py:120-234`` (pass-2 query builder + edge vote) in numpy/scipy.
``rtree`` is absent here; a point-in-closed-box filter is result-identical to
``rtree.intersection`` for degenerate point boxes (ids come back in a different
order, which only permutes patch-local indices; edge votes are keyed by global
index so the result is unchanged — we sort ids to make it deterministic).
"""
from collections import defaultdict

import numpy as np
import scipy.spatial
import torch


def get_patch_info_one_img(image_index, image_size, sample_margin, patch_size, patches_per_edge):
    """dataset.py:56-67 — x outer / y inner, python round() (banker's)."""
    smin = sample_margin
    smax = image_size - (patch_size + sample_margin)
    samples = [round(x) for x in np.linspace(start=smin, stop=smax, num=patches_per_edge)]
    return [(image_index, (x, y), (x + patch_size, y + patch_size))
            for x in samples for y in samples]


def get_batch_img_patches(img, batch_patch_info):
    """inferencer.py:52-58."""
    return torch.stack([torch.tensor(img[y0:y1, x0:x1, :], dtype=torch.float32)
                        for _, (x0, y0), (x1, y1) in batch_patch_info], 0).contiguous()


def fuse_masks(image_hw, patch_infos, mask_scores_list):
    """inferencer.py:79-110: scatter-add, divide by coverage count, x255, trunc
    to u8.  Uncovered border is 0/0 = NaN whose u8 cast is 0 (probed on CPU torch,
    SURVEY App. D.2) — made explicit here."""
    kp = torch.zeros(image_hw, dtype=torch.float32)
    road = torch.zeros(image_hw, dtype=torch.float32)
    cnt = torch.zeros(image_hw, dtype=torch.float32)
    i = 0
    for scores in mask_scores_list:
        for j in range(scores.shape[0]):
            _, (x0, y0), (x1, y1) = patch_infos[i]
            kp[y0:y1, x0:x1] += scores[j, :, :, 0]
            road[y0:y1, x0:x1] += scores[j, :, :, 1]
            cnt[y0:y1, x0:x1] += 1.0
            i += 1
    kp, road = kp / cnt, road / cnt
    to_u8 = lambda t: torch.nan_to_num(t * 255, nan=0.0).to(torch.uint8).numpy()
    return to_u8(kp), to_u8(road)


def nms_points(points, scores, radius):
    """graph_utils.py:572-591 — greedy radius NMS in descending-score order
    (np.argsort()[::-1] tie order), score > 1.0 is force-kept."""
    order = np.argsort(scores)[::-1]
    pts = points[order, :]
    sc = scores[order]
    kept = np.ones(order.shape[0], dtype=bool)
    if pts.shape[0] == 0:
        return pts
    tree = scipy.spatial.KDTree(pts)
    for i, p in enumerate(pts):
        if not kept[i]:
            continue
        nbr = tree.query_ball_point(p, r=radius)
        kept[nbr] = np.greater(sc[nbr], 1.0)
        kept[i] = True
    return pts[kept]


def _points_and_scores(mask, thr):
    """graph_extraction.py:24-28 — (x, y) order."""
    rcs = np.column_stack(np.where(mask > thr))
    return rcs[:, ::-1], mask[mask > thr]


def extract_graph_points(keypoint_mask, road_mask, config):
    """graph_extraction.py:130-139."""
    c, s = _points_and_scores(keypoint_mask, config.ITSC_THRESHOLD * 255)
    k0 = nms_points(c, s, config.ITSC_NMS_RADIUS)
    c, s = _points_and_scores(road_mask, config.ROAD_THRESHOLD * 255)
    k1 = nms_points(c, s, config.ROAD_NMS_RADIUS)
    c = np.concatenate([k0, k1], axis=0)
    s = np.concatenate([np.ones(k0.shape[0]), np.zeros(k1.shape[0])], axis=0)
    return nms_points(c, s, config.ROAD_NMS_RADIUS)


def build_patch_queries(graph_points, patch_info, config):
    """inferencer.py:148-176 for one tile."""
    _, (x0, y0), (x1, y1) = patch_info
    gx, gy = graph_points[:, 0], graph_points[:, 1]
    ids = np.nonzero((gx >= x0) & (gx <= x1) & (gy >= y0) & (gy <= y1))[0]
    n = len(ids)
    k = config.MAX_NEIGHBOR_QUERIES
    pts = graph_points[ids, :] - np.array([[x0, y0]], dtype=graph_points.dtype)
    if n == 0:
        return (ids, pts.reshape(0, 2), np.zeros((0, k, 2), np.int64), np.zeros((0, k), bool))
    tree = scipy.spatial.KDTree(pts)
    _, knn = tree.query(pts, k=k + 1, distance_upper_bound=config.NEIGHBOR_RADIUS)
    knn = knn[:, 1:]
    src = np.tile(np.arange(n)[:, None], (1, k))
    valid = knn < n
    tgt = np.where(valid, knn, src)
    return ids, pts, np.stack([src, tgt], -1), valid


def collate(x_list):
    """inferencer.py:179-185."""
    length = max(x.shape[0] for x in x_list)
    return np.stack([np.pad(x, [(0, length - x.shape[0])] + [(0, 0)] * (x.ndim - 1))
                     for x in x_list], 0)


def vote_edges(votes, threshold):
    """inferencer.py:224-228: directed mean score > threshold."""
    sums, cnts = votes
    return np.array([e for e, s in sums.items() if s / cnts[e] > threshold]).reshape(-1, 2)


def infer_pass1(net, img, config):
    """inferencer.py:61-110: tile batches -> masks + cached features -> fused u8 masks."""
    size = img.shape[0]
    bs = config.INFER_BATCH_SIZE
    infos = get_patch_info_one_img(0, size, config.SAMPLE_MARGIN, config.PATCH_SIZE,
                                   config.INFER_PATCHES_PER_EDGE)
    nb = (len(infos) + bs - 1) // bs
    feats, scores = [], []
    for bi in range(nb):
        batch = get_batch_img_patches(img, infos[bi * bs:(bi + 1) * bs])
        s, f = net.infer_masks_and_img_features(batch)
        feats.append(f)
        scores.append(s)
    kp_mask, road_mask = fuse_masks(img.shape[:2], infos, scores)
    return infos, feats, kp_mask, road_mask


def infer_pass2(net, feats, points, infos, config):
    """inferencer.py:120-234: per-tile queries -> TopoNet -> directed edge votes -> thresholded edges."""
    bs = config.INFER_BATCH_SIZE
    nb = (len(infos) + bs - 1) // bs
    sums, cnts = defaultdict(float), defaultdict(float)
    for bi in range(nb):
        qs = [build_patch_queries(points, pi, config) for pi in infos[bi * bs:(bi + 1) * bs]]
        pts = collate([q[1] for q in qs])
        pairs = collate([q[2] for q in qs])
        valid = collate([q[3] for q in qs])
        if pts.shape[1] == 0:
            continue
        ts = net.infer_toponet(feats[bi], torch.tensor(pts), torch.tensor(pairs), torch.tensor(valid))
        ts = torch.where(torch.isnan(ts), -100.0, ts).squeeze(-1).numpy()
        for b, q in enumerate(qs):
            ids = q[0]
            for si in range(len(ids)):
                for pi in range(pairs.shape[2]):
                    if not valid[b, si, pi]:
                        continue
                    e = (int(ids[pairs[b, si, pi, 0]]), int(ids[pairs[b, si, pi, 1]]))
                    sums[e] += float(ts[b, si, pi])
                    cnts[e] += 1.0
    return vote_edges((sums, cnts), config.TOPO_THRESHOLD), sums, cnts


def infer_one_img(net, img, config):
    """inferencer.py:61-234 end-to-end on the CPU oracle ``net``."""
    infos, feats, kp_mask, road_mask = infer_pass1(net, img, config)
    points = extract_graph_points(kp_mask, road_mask, config)
    if points.shape[0] == 0:
        return points, np.zeros((0, 2), np.int32), kp_mask, road_mask
    edges, _, _ = infer_pass2(net, feats, points, infos, config)
    return points[:, ::-1], edges, kp_mask, road_mask
