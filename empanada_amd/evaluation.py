"""Panoptic quality between two labelled volumes (the "PQ vs CPU ref" half of the headline metric).

Formula: empanada/evaluation/panoptic_metrics.py:3-54; matching: Hungarian on the instance IoU matrix with the
matches kept at IoU >= 0.5 (empanada/evaluation/evaluator.py:88-89 -> rle_matcher, inference/matcher.py:136-232).
The IoU table comes from one joint histogram of (gt label, pred label) over the voxels.
"""
import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

__all__ = ['panoptic_quality', 'volume_pq', 'f1', 'ap', 'precision', 'recall', 'f1_50', 'f1_75', 'precision_50',
           'precision_75', 'recall_50', 'recall_75', 'iou', 'Evaluator']


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


# ----------------------------------------------------------------------------- detection metrics
def _counts(gt_unmatched, pred_unmatched, matched_ious, iou_thr):
    matched_ious = np.asarray(matched_ious, dtype=float)
    tp = int(np.count_nonzero(matched_ious >= iou_thr))
    failed = int(np.count_nonzero(matched_ious < iou_thr))       # a failed match costs one fp and one fn
    return tp, len(pred_unmatched) + failed, len(gt_unmatched) + failed


def f1(gt_matched, gt_unmatched, pred_matched, pred_unmatched, matched_ious, iou_thr=0.5):
    """instance_metrics.py:3-54 (1 for two empty masks)"""
    tp, fp, fn = _counts(gt_unmatched, pred_unmatched, matched_ious, iou_thr)
    return 1 if tp + fp + fn == 0 else tp / (tp + 0.5 * fp + 0.5 * fn)


def ap(gt_matched, gt_unmatched, pred_matched, pred_unmatched, matched_ious, iou_thr=0.5):
    """instance_metrics.py:56-108"""
    tp, fp, fn = _counts(gt_unmatched, pred_unmatched, matched_ious, iou_thr)
    return 1 if tp + fp + fn == 0 else tp / (tp + fp + fn)


def precision(gt_matched, gt_unmatched, pred_matched, pred_unmatched, matched_ious, iou_thr=0.5):
    """instance_metrics.py:110-156"""
    tp, fp, _ = _counts(gt_unmatched, pred_unmatched, matched_ious, iou_thr)
    return 1 if tp + fp == 0 else tp / (tp + fp)


def recall(gt_matched, gt_unmatched, pred_matched, pred_unmatched, matched_ious, iou_thr=0.5):
    """instance_metrics.py:158-206"""
    tp, _, fn = _counts(gt_unmatched, pred_unmatched, matched_ious, iou_thr)
    return 1 if tp + fn == 0 else tp / (tp + fn)


def f1_50(**kw): return f1(**kw, iou_thr=0.5)
def f1_75(**kw): return f1(**kw, iou_thr=0.75)
def precision_50(**kw): return precision(**kw, iou_thr=0.5)
def precision_75(**kw): return precision(**kw, iou_thr=0.75)
def recall_50(**kw): return recall(**kw, iou_thr=0.5)
def recall_75(**kw): return recall(**kw, iou_thr=0.75)


def iou(gt_rle, pred_rle):
    """semantic_metrics.py:4-26: IoU of two (n, 2) (start, run) tables."""
    from .array_utils import rle_iou
    if len(gt_rle) == 0 and len(pred_rle) == 0:
        return 1
    if len(gt_rle) == 0 or len(pred_rle) == 0:
        return 0
    return rle_iou(gt_rle[:, 0], gt_rle[:, 1], pred_rle[:, 0], pred_rle[:, 1])


class Evaluator:
    """evaluator.py:24-122: scores a predicted tracker json against a ground-truth tracker json (the files
    InstanceTracker.write_to_json produces).  Matching runs through the product rle_matcher (box screening and
    run intersections in libemp_hip.so).

    Deviation, on purpose: the reference hands the json's {'box', 'rle'} entries straight to rle_matcher, which
    reads 'starts'/'runs' and therefore raises KeyError on any non-empty file (evaluator.py:88-89 with
    matcher.py:104-118); here the rle strings are decoded first."""

    def __init__(self, semantic_metrics=None, instance_metrics=None, panoptic_metrics=None):
        self.semantic_metrics = semantic_metrics
        self.instance_metrics = instance_metrics
        self.panoptic_metrics = panoptic_metrics

    @staticmethod
    def _unpack_instance_dict(instance_dict):
        labels, boxes, encodings = [], [], []
        for k, v in instance_dict.items():
            labels.append(int(k))
            boxes.append(v['box'])
            encodings.append(v['rle'])
        return np.array(labels), np.array(boxes), encodings

    @staticmethod
    def _decoded(instance_dict):
        from .array_utils import string_to_rle
        out = {}
        for k, v in instance_dict.items():
            starts, runs = string_to_rle(v['rle'])
            out[int(k)] = {'box': tuple(v['box']), 'starts': starts, 'runs': runs}
        return out

    def __call__(self, gt_json_fpath, pred_json_fpath, return_instances=False):
        import json
        from .array_utils import merge_rles, string_to_rle
        from .inference.matcher import rle_matcher
        with open(gt_json_fpath) as f:
            gt_json = json.load(f)
        with open(pred_json_fpath) as f:
            pred_json = json.load(f)
        assert gt_json['class_id'] == pred_json['class_id'], "Prediction and ground truth classes must match!"

        _, _, gt_enc = self._unpack_instance_dict(gt_json['instances'])
        _, _, pred_enc = self._unpack_instance_dict(pred_json['instances'])
        semantic_results, instance_results, panoptic_results = {}, {}, {}

        if self.semantic_metrics is not None:
            gt_rle = np.concatenate([np.stack(string_to_rle(e), axis=1) for e in gt_enc])
            if len(pred_enc) > 1:                        # evaluator.py:6-22 (a single prediction scores as [-1, -1])
                pr = np.concatenate([np.stack(string_to_rle(e), axis=1) for e in pred_enc])
                pred_rle = np.stack(merge_rles(pr[:, 0], pr[:, 1]), axis=1)
            else:
                pred_rle = np.array([[-1, -1]])
            semantic_results = {n: fn(gt_rle, pred_rle) for n, fn in self.semantic_metrics.items()}

        gt_matched = pred_matched = gt_unmatched = pred_unmatched = matched_ious = None
        if self.instance_metrics is not None or self.panoptic_metrics is not None:
            matched, all_labels, matched_ious = rle_matcher(self._decoded(gt_json['instances']),
                                                            self._decoded(pred_json['instances']))
            gt_matched, pred_matched = matched
            gt_unmatched = np.setdiff1d(all_labels[0], gt_matched)
            pred_unmatched = np.setdiff1d(all_labels[1], pred_matched)
            kw = dict(gt_matched=gt_matched, pred_matched=pred_matched, gt_unmatched=gt_unmatched,
                      pred_unmatched=pred_unmatched, matched_ious=matched_ious)
            if self.instance_metrics is not None:
                instance_results = {n: fn(**kw) for n, fn in self.instance_metrics.items()}
            if self.panoptic_metrics is not None:
                panoptic_results = {n: fn(**kw) for n, fn in self.panoptic_metrics.items()}

        results = {**semantic_results, **instance_results, **panoptic_results}
        if return_instances:
            return results, dict(gt_matched=gt_matched, pred_matched=pred_matched, gt_unmatched=gt_unmatched,
                                 pred_unmatched=pred_unmatched, matched_ious=matched_ious)
        return results
