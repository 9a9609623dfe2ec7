"""Instance matching across consecutive slices, reference names and semantics
(``empanada/inference/matcher.py``): ``rle_matcher`` :136-232, ``RLEMatcher`` :234-326,
``merge_attrs`` :14-28, plus ``SequentialMatcher`` which scripts/inference3d_multigpu.py:25,340-343
imports but the reference never defines.

All run-length intersections come from libemp_hip.so (one emp_rle_pair_intersections launch per
call, over every box-screened pair); the Hungarian assignment is scipy's, exactly the third-party
routine the reference calls (matcher.py:213), so that tie-breaking is identical.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment

from ..array_utils import box_pairs, merge_boxes, merge_rles, rle_pair_intersections
from .deferred import _Pending
from .rle import pan_seg_to_rle_seg, rle_seg_to_pan_seg, unpack_rle_attrs

__all__ = ['rle_matcher', 'RLEMatcher', 'SequentialMatcher', 'merge_attrs']


def merge_attrs(rle_attr1, rle_attr2):
    """matcher.py:14-28"""
    starts, runs = merge_rles(rle_attr1['starts'], rle_attr1['runs'], rle_attr2['starts'], rle_attr2['runs'])
    return {'box': merge_boxes(rle_attr1['box'], rle_attr2['box']), 'starts': starts, 'runs': runs}


def rle_matcher(target_instance_rles, match_instance_rles, iou_thr=0.5, return_iou=False, return_ioa=False):
    """matcher.py:136-232"""
    t_labels, t_boxes, t_starts, t_runs = unpack_rle_attrs(target_instance_rles)
    m_labels, m_boxes, m_starts, m_runs = unpack_rle_attrs(match_instance_rles)
    if len(t_labels) == 0 or len(m_labels) == 0:
        empty = np.array([])
        if return_ioa:
            return (empty, empty), (t_labels, m_labels), empty, empty
        return (empty, empty), (t_labels, m_labels), empty

    iou = np.zeros((len(t_boxes), len(m_boxes)), dtype='float')
    ioa = np.zeros((len(t_boxes), len(m_boxes)), dtype=np.float32)
    rows, cols, _, _ = box_pairs(t_boxes, m_boxes)
    if len(rows):
        nt = len(t_labels)
        inter = rle_pair_intersections(list(t_starts) + list(m_starts), list(t_runs) + list(m_runs),
                                       np.stack([rows, cols + nt], axis=1))
        t_area = np.array([int(np.sum(r)) for r in t_runs], dtype=np.int64)
        m_area = np.array([int(np.sum(r)) for r in m_runs], dtype=np.int64)
        iou[rows, cols] = inter / (t_area[rows] + m_area[cols] - inter)      # int64 / int64 -> fp64
        ioa[rows, cols] = inter / m_area[cols]                                # stored as fp32 (:196)

    match_rows, match_cols = linear_sum_assignment(iou, maximize=True)
    if iou_thr is not None:
        keep = iou[match_rows, match_cols] >= iou_thr
        match_rows, match_cols = match_rows[keep], match_cols[keep]
    output = ((t_labels[match_rows], m_labels[match_cols]), [t_labels, m_labels], iou[(match_rows, match_cols)])
    if return_iou:
        output = output + (iou,)
    if return_ioa:
        output = output + (ioa,)
    return output


def assign_labels(match_labels, target_labels, matched, ioa, merge_ioa_thr, assign_new, next_label):
    """Label propagation rule of RLEMatcher.__call__ (matcher.py:292-319), shared by the RLE and the
    table-based matchers.  Returns (new label per match instance, next_label)."""
    label_matches = {ml: tl for tl, ml in zip(matched[0], matched[1])}
    out = []
    for i, ml in enumerate(match_labels):
        if ml in label_matches:
            new_label = label_matches[ml]
        else:
            ioa_max = ioa[:, i].max() if len(ioa) > 0 else 0
            if ioa_max >= merge_ioa_thr:
                new_label = target_labels[ioa[:, i].argmax()]
            elif assign_new:
                new_label = next_label
                next_label += 1
            else:
                new_label = ml
        out.append(new_label)
    return out, next_label


class RLEMatcher:
    """matcher.py:234-326"""

    def __init__(self, class_id, label_divisor, merge_iou_thr=0.25, merge_ioa_thr=0.25, assign_new=True, **kwargs):
        self.class_id = class_id
        self.label_divisor = label_divisor
        self.merge_iou_thr = merge_iou_thr
        self.merge_ioa_thr = merge_ioa_thr
        self.assign_new = assign_new
        self.next_label = (class_id * label_divisor) + 1
        self.target_rle = None

    def __getstate__(self):
        """pickling (matchers cross mp.Queues in the reference's scripts): a matcher whose slices are deferred
        (inference/deferred.py) files them first, so that target AND label counter are the real ones"""
        if isinstance(self.target_rle, _Pending):
            self.target_rle.resolve()
        return self.__dict__.copy()

    def initialize_target(self, target_instance_rles):
        self.target_rle = target_instance_rles
        objs = list(target_instance_rles.keys())
        if len(objs) > 0:
            self.next_label = max(objs) + 1

    def update_target(self, instance_rles):
        self.target_rle = instance_rles

    def __call__(self, match_instance_rle, update_target=True):
        assert self.target_rle is not None, "Initialize target rle before running!"
        matched_labels, all_labels, _, ioa_matrix = rle_matcher(
            self.target_rle, match_instance_rle, self.merge_iou_thr, return_ioa=True)
        target_labels, match_labels = all_labels
        assert list(match_labels) == [int(k) for k in match_instance_rle.keys()]
        new_labels, self.next_label = assign_labels(
            list(match_instance_rle.keys()), target_labels, matched_labels, ioa_matrix, self.merge_ioa_thr,
            self.assign_new, self.next_label)
        matched_rles = {}
        for new_label, mattrs in zip(new_labels, match_instance_rle.values()):
            if new_label not in matched_rles:
                matched_rles[new_label] = mattrs
            else:
                matched_rles[new_label] = merge_attrs(matched_rles[new_label], mattrs)
        if update_target:
            self.update_target(matched_rles)
        return matched_rles


class SequentialMatcher:
    """Dense-image matcher with the interface scripts/inference3d_multigpu.py:340-343,504-505 expects:
    ``target_seg`` (None until initialised), ``initialize_target(pan_seg) -> pan_seg``,
    ``__call__(pan_seg) -> pan_seg``, attributes ``assign_new`` and ``force_connected``.
    It is RLEMatcher applied to one thing class of a dense panoptic image."""

    def __init__(self, thing_class, label_divisor, merge_iou_thr=0.25, merge_ioa_thr=0.25, assign_new=True,
                 force_connected=True, **kwargs):
        self.thing_class = thing_class
        self.label_divisor = label_divisor
        self.force_connected = force_connected
        self._m = RLEMatcher(thing_class, label_divisor, merge_iou_thr, merge_ioa_thr, assign_new)
        self.target_seg = None

    @property
    def assign_new(self):
        return self._m.assign_new

    @assign_new.setter
    def assign_new(self, v):
        self._m.assign_new = v

    def _rle(self, pan_seg):
        pan = np.asarray(pan_seg)
        return pan, pan_seg_to_rle_seg(pan, [self.thing_class], self.label_divisor, [self.thing_class],
                                       self.force_connected)

    def _paint(self, pan, rle_seg):
        lo = self.thing_class * self.label_divisor
        out = pan.copy()
        out[(pan >= lo) & (pan < lo + self.label_divisor)] = 0
        painted = rle_seg_to_pan_seg(rle_seg, pan.shape).astype(out.dtype)
        return np.where(painted > 0, painted, out)

    def initialize_target(self, pan_seg):
        pan, rle_seg = self._rle(pan_seg)
        self._m.target_rle = None
        self._m.initialize_target(rle_seg[self.thing_class])
        self.target_seg = self._paint(pan, rle_seg)
        return self.target_seg

    def __call__(self, pan_seg):
        assert self.target_seg is not None, "Initialize target seg before running!"
        pan, rle_seg = self._rle(pan_seg)
        rle_seg[self.thing_class] = self._m(rle_seg[self.thing_class])
        self.target_seg = self._paint(pan, rle_seg)
        return self.target_seg
