"""Panoptic quality between two labelled volumes (the "PQ vs CPU ref" half of the headline metric).

Formula: empanada/evaluation/panoptic_metrics.py:3-54; matching: Hungarian on the instance IoU matrix with the
matches kept at IoU >= 0.5 (empanada/evaluation/evaluator.py:88-89 -> rle_matcher, inference/matcher.py:136-232).
The IoU table comes from one joint histogram of (gt label, pred label) over the voxels.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

__all__ = ['panoptic_quality', 'volume_pq']


def panoptic_quality(gt_matched, gt_unmatched, pred_matched, pred_unmatched, matched_ious):
    """panoptic_metrics.py:3-54"""
    matched_ious = np.asarray(matched_ious, dtype=float)
    fn = len(gt_unmatched)
    fp = len(pred_unmatched)
    tp_ious = matched_ious[matched_ious >= 0.5]
    tp = len(tp_ious)
    failed = np.count_nonzero(matched_ious < 0.5)
    fp += failed
    fn += failed
    if tp + fp + fn == 0:
        return 1
    sq = tp_ious.sum() / (tp + 1e-5)
    rq = tp / (tp + 0.5 * fp + 0.5 * fn)
    return sq * rq


def volume_pq(gt, pred, iou_thr=0.5):
    """PQ of `pred` against `gt` (integer label volumes of equal shape, 0 = background; numpy or torch).
    Returns (pq, n_gt, n_pred, n_matched)."""
    g = torch.as_tensor(np.asarray(gt).astype(np.int64) if not isinstance(gt, torch.Tensor) else gt).reshape(-1).long()
    p = torch.as_tensor(np.asarray(pred).astype(np.int64) if not isinstance(pred, torch.Tensor) else pred).reshape(-1).long()
    if p.device != g.device:
        p = p.to(g.device)
    gl, gi = torch.unique(g, return_inverse=True)
    pl, pi = torch.unique(p, return_inverse=True)
    joint = torch.bincount(gi * len(pl) + pi, minlength=len(gl) * len(pl)).reshape(len(gl), len(pl)).cpu().numpy()
    gl, pl = gl.cpu().numpy(), pl.cpu().numpy()
    gk, pk = gl != 0, pl != 0
    inter = joint[gk][:, pk].astype(np.float64)
    ga = joint[gk].sum(axis=1).astype(np.float64)
    pa = joint[:, pk].sum(axis=0).astype(np.float64)
    gl, pl = gl[gk], pl[pk]
    if len(gl) == 0 or len(pl) == 0:
        return panoptic_quality([], gl, [], pl, []), len(gl), len(pl), 0
    iou = inter / (ga[:, None] + pa[None, :] - inter)
    rows, cols = linear_sum_assignment(iou, maximize=True)
    keep = iou[rows, cols] >= iou_thr
    rows, cols = rows[keep], cols[keep]
    pq = panoptic_quality(gl[rows], np.setdiff1d(gl, gl[rows]), pl[cols], np.setdiff1d(pl, pl[cols]), iou[rows, cols])
    return float(pq), len(gl), len(pl), int(len(rows))
