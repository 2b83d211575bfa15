"""Mask -> graph points (host side, between the two GPU passes): threshold + greedy radius NMS.

Mirrors reference graph_extraction.py:24-28,130-139 and graph_utils.py:572-591.  This step is a
"next" row of SURVEY.md §8(f) (rank 2): it stays on the host as in the reference; the greedy NMS is
order-dependent, so the candidate order (descending score, np.argsort tie order) is kept identical.
"""
import numpy as np
import scipy.spatial


def points_and_scores_from_mask(mask, threshold):
    sel = mask > threshold
    rc = np.column_stack(np.where(sel))
    return rc[:, ::-1], mask[sel]          # (x, y), scores


def nms_points(points, scores, radius, return_indices=False):
    """Greedy radius suppression in descending score order; a score > 1.0 is always kept."""
    order = np.argsort(scores)[::-1]
    pts, sc = points[order, :], scores[order]
    kept = np.ones(order.shape[0], dtype=bool)
    if pts.shape[0]:
        tree = scipy.spatial.cKDTree(pts)
        force = sc > 1.0
        for i in range(pts.shape[0]):
            if not kept[i]:
                continue
            nbr = tree.query_ball_point(pts[i], r=radius)
            kept[nbr] = force[nbr]
            kept[i] = True
    if return_indices:
        return pts[kept], order[kept]
    return pts[kept]


def extract_graph_points(keypoint_mask, road_mask, config):
    cand, sc = points_and_scores_from_mask(keypoint_mask, config.ITSC_THRESHOLD * 255)
    kp0 = nms_points(cand, sc, config.ITSC_NMS_RADIUS)
    cand, sc = points_and_scores_from_mask(road_mask, config.ROAD_THRESHOLD * 255)
    kp1 = nms_points(cand, sc, config.ROAD_NMS_RADIUS)
    cand = np.concatenate([kp0, kp1], axis=0)
    prio = np.concatenate([np.ones(kp0.shape[0]), np.zeros(kp1.shape[0])], axis=0)  # intersections first
    return nms_points(cand, prio, config.ROAD_NMS_RADIUS)
