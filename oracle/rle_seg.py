"""CPU oracle for pan_seg <-> rle_seg, matching, tracking, filters (TEST INFRASTRUCTURE ONLY).

numpy restatement of empanada/inference/{rle,matcher,tracker,filters}.py and the
matching half of empanada/inference/patterns.py.
"""
import ctypes
import math

import numpy as np
from scipy.optimize import linear_sum_assignment

from ._clib import lib
from .rle_ops import (box_pairs, merge_boxes, merge_rles, rle_decode, rle_encode, rle_ioa, rle_iou,
                      string_to_rle)

_i64p = ctypes.POINTER(ctypes.c_int64)


# ----------------------------------------------------------------------------- rle.py
def connected_components(seg):
    """rle.py:18-24 -- third-party contract (cc3d connectivity=8 / skimage.measure.label):
    multi-value 8-connected labelling, ids 1..n in raster order of first pixel.  PARITY UNPINNED
    (neither library is in the image; no reference test runs force_connected=True)."""
    seg = np.ascontiguousarray(seg, dtype=np.int64)
    out = np.empty_like(seg)
    h, w = seg.shape
    lib().emp_oracle_cc8(seg.ctypes.data_as(_i64p), h, w, out.ctypes.data_as(_i64p))
    return out


def _regionprops(label_img):
    """skimage.measure.regionprops contract used at rle.py:75-81: ascending label,
    bbox (y0,x0,y1,x1) half-open, coords row-major.  Yields (label, bbox, flat_indices)."""
    flat = label_img.ravel()
    idx = np.flatnonzero(flat)
    if idx.size == 0:
        return
    order = np.argsort(flat[idx], kind='stable')       # group by label, raster order inside
    idx = idx[order]
    labs = flat[idx]
    cuts = np.flatnonzero(np.diff(labs)) + 1
    w = label_img.shape[1]
    for chunk in np.split(idx, cuts):
        ys, xs = chunk // w, chunk % w
        bbox = (int(ys.min()), int(xs.min()), int(ys.max()) + 1, int(xs.max()) + 1)
        yield int(flat[chunk[0]]), bbox, chunk.astype(np.int64)


def pan_seg_to_rle_seg(pan_seg, labels, label_divisor, thing_list, force_connected=True):
    """rle.py:26-86"""
    pan_seg = np.asarray(pan_seg)
    rle_seg = {}
    for label in labels:
        min_id = label * label_divisor
        max_id = min_id + label_divisor
        inst = pan_seg.astype(np.int64).copy()
        inst[(pan_seg < min_id) | (pan_seg >= max_id)] = 0
        if force_connected and label in thing_list:
            inst = connected_components(inst)
            inst[inst > 0] += min_id
        attrs = {}
        for lab, bbox, coords_flat in _regionprops(inst):
            starts, runs = rle_encode(coords_flat)
            attrs[lab] = {'box': bbox, 'starts': starts, 'runs': runs}
        rle_seg[label] = attrs
    return rle_seg


def rle_seg_to_pan_seg(rle_seg, shape):
    """rle.py:88-118"""
    pan = np.zeros(shape, dtype=np.uint32).ravel()
    for attrs in rle_seg.values():
        for object_id, a in attrs.items():
            for s, r in zip(a['starts'], a['runs']):
                pan[s:s + r] = object_id
    return pan.reshape(shape)


def unpack_rle_attrs(instance_rle_seg):
    """rle.py:120-150"""
    labels, boxes, starts, runs = [], [], [], []
    for label, attrs in instance_rle_seg.items():
        labels.append(int(label))
        boxes.append(attrs['box'])
        if 'rle' in attrs:
            s, r = string_to_rle(attrs['rle'])
        else:
            s, r = attrs['starts'], attrs['runs']
        starts.append(s)
        runs.append(r)
    return np.array(labels), np.array(boxes), starts, runs


# ----------------------------------------------------------------------------- matcher.py
def merge_attrs(a1, a2):
    """matcher.py:14-28"""
    starts, runs = merge_rles(a1['starts'], a1['runs'], a2['starts'], a2['runs'])
    return {'box': merge_boxes(a1['box'], a2['box']), 'starts': starts, 'runs': runs}


def rle_matcher(target_rles, match_rles, iou_thr=0.5, return_iou=False, return_ioa=False):
    """matcher.py:136-232 -- box screen, RLE IoU (fp64) / IoA (fp32 matrix), Hungarian, IoU filter."""
    t_labels, t_boxes, t_starts, t_runs = unpack_rle_attrs(target_rles)
    m_labels, m_boxes, m_starts, m_runs = unpack_rle_attrs(match_rles)
    if len(t_labels) == 0 or len(m_labels) == 0:
        empty = np.array([])
        if return_ioa:
            return (empty, empty), (t_labels, m_labels), empty, empty
        return (empty, empty), (t_labels, m_labels), empty

    iou = np.zeros((len(t_boxes), len(m_boxes)), dtype='float')
    ioa = np.zeros((len(t_boxes), len(m_boxes)), dtype=np.float32)
    rows, cols, _, _ = box_pairs(t_boxes, m_boxes)
    for r1, r2 in zip(rows, cols):
        iou[r1, r2] = rle_iou(t_starts[r1], t_runs[r1], m_starts[r2], m_runs[r2])
        ioa[r1, r2] = rle_ioa(t_starts[r1], t_runs[r1], m_starts[r2], m_runs[r2])

    mr, mc = linear_sum_assignment(iou, maximize=True)
    if iou_thr is not None:
        keep = iou[mr, mc] >= iou_thr
        mr, mc = mr[keep], mc[keep]
    out = ((t_labels[mr], m_labels[mc]), [t_labels, m_labels], iou[(mr, mc)])
    if return_iou:
        out = out + (iou,)
    if return_ioa:
        out = out + (ioa,)
    return out


class RLEMatcher:
    """matcher.py:234-326"""

    def __init__(self, class_id, label_divisor, merge_iou_thr=0.25, merge_ioa_thr=0.25,
                 assign_new=True, **kwargs):
        self.class_id = class_id
        self.label_divisor = label_divisor
        self.merge_iou_thr = merge_iou_thr
        self.merge_ioa_thr = merge_ioa_thr
        self.assign_new = assign_new
        self.next_label = (class_id * label_divisor) + 1
        self.target_rle = None

    def initialize_target(self, target_instance_rles):
        self.target_rle = target_instance_rles
        objs = list(target_instance_rles.keys())
        if len(objs) > 0:
            self.next_label = max(objs) + 1

    def update_target(self, instance_rles):
        self.target_rle = instance_rles

    def __call__(self, match_instance_rle, update_target=True):
        assert self.target_rle is not None, "Initialize target rle before running!"
        matched_labels, all_labels, _, ioa = rle_matcher(
            self.target_rle, match_instance_rle, self.merge_iou_thr, return_ioa=True)
        target_labels, match_labels = all_labels
        label_matches = {ml: tl for tl, ml in zip(matched_labels[0], matched_labels[1])}

        matched = {}
        for i, (ml, mattrs) in enumerate(match_instance_rle.items()):
            if ml in label_matches:
                new_label = label_matches[ml]
            else:
                assert ml == match_labels[i]
                ioa_max = ioa[:, i].max() if len(ioa) > 0 else 0
                if ioa_max >= self.merge_ioa_thr:
                    new_label = target_labels[ioa[:, i].argmax()]
                elif self.assign_new:
                    new_label = self.next_label
                    self.next_label += 1
                else:
                    new_label = ml
            if new_label not in matched:
                matched[new_label] = mattrs
            else:
                matched[new_label] = merge_attrs(matched[new_label], mattrs)
        if update_target:
            self.update_target(matched)
        return matched


# ----------------------------------------------------------------------------- patterns.py (matching)
def create_matchers(thing_list, label_divisor, merge_iou_thr, merge_ioa_thr):
    """patterns.py:33-39"""
    return [RLEMatcher(c, label_divisor, merge_iou_thr, merge_ioa_thr) for c in thing_list]


def apply_matchers(rle_seg, matchers):
    """patterns.py:55-66"""
    for m in matchers:
        if m.target_rle is None:
            m.initialize_target(rle_seg[m.class_id])
        else:
            rle_seg[m.class_id] = m(rle_seg[m.class_id])
    return rle_seg


def forward_matching(pan_segs, matchers, labels, label_divisor, thing_list):
    """patterns.py:68-100 without the mp.Queue: pan_seg -> rle_seg -> apply_matchers, per slice."""
    rle_stack = []
    for pan in pan_segs:
        if pan is None:
            continue
        rle_seg = pan_seg_to_rle_seg(pan, labels, label_divisor, thing_list, force_connected=True)
        rle_stack.append(apply_matchers(rle_seg, matchers))
    return rle_stack


def forward_multigpu(items, matchers, confidence_thr, median_kernel_size, labels, label_divisor, thing_list,
                     stuff_area=32, void_label=0):
    """patterns.py:279-350 without the mp.Queue / Pipe: items = [(sem probabilities (1,C,H,W), instance cells
    (1,1,H,W)), ...] in arrival order; median queue over 'sem' -> harden -> get_panoptic_seg -> rle_seg ->
    apply_matchers.  Returns the rle_stack the reference sends through the pipe."""
    from . import postprocess as OP
    q = OP.MedianQueue(median_kernel_size)
    rle_stack = []

    def consume(o):
        sem = OP.harden_seg(o['sem'], confidence_thr)[0]
        pan = OP.get_panoptic_seg(sem, np.asarray(o['cells'], dtype=np.float32), label_divisor, thing_list,
                                  stuff_area, void_label)
        rle_seg = pan_seg_to_rle_seg(pan.squeeze(), labels, label_divisor, thing_list, force_connected=True)
        rle_stack.append(apply_matchers(rle_seg, matchers))

    for sem, cells in items:
        q.enqueue({'sem': np.asarray(sem, dtype=np.float32), 'cells': cells})
        o = q.get_next(['sem'])
        if o is not None:
            consume(o)
    for o in q.end():
        consume(o)
    return rle_stack


def backward_matching(rle_stack, matchers, axis_len):
    """patterns.py:102-121 -- generator over slices n-1..0, assign_new=False, stack mutated in place."""
    for m in matchers:
        m.target_rle = None
        m.assign_new = False
    for rev_idx in range(axis_len - 1, -1, -1):
        rle_seg = apply_matchers(rle_stack[rev_idx], matchers)
        yield rev_idx, rle_seg


# ----------------------------------------------------------------------------- tracker.py
_AXIS_NUM = {'xy': 0, 'xz': 1, 'yz': 2}


def to_box3d(index2d, box, axis):
    """tracker.py:11-23"""
    h1, w1, h2, w2 = box
    if axis == 'xy':
        return (index2d, h1, w1, index2d + 1, h2, w2)
    if axis == 'xz':
        return (h1, index2d, w1, h2, index2d + 1, w2)
    return (h1, w1, index2d, h2, w2, index2d + 1)


class InstanceTracker:
    """tracker.py:40-159 (JSON (de)serialisation omitted: wire format)."""

    def __init__(self, class_id=None, label_divisor=None, shape3d=None, axis='xy'):
        assert axis in ('xy', 'xz', 'yz')
        self.class_id = class_id
        self.label_divisor = label_divisor
        self.shape3d = shape3d
        self.axis = axis
        self.finished = False
        self.instances = {}

    def update(self, instance_rles, index2d):
        """tracker.py:61-100.  xz maps only the run *starts* to 3D (row-wrap bug reproduced)."""
        assert not self.finished, "Cannot update tracker after calling finish!"
        ignore = _AXIS_NUM[self.axis]
        shape2d = tuple(s for i, s in enumerate(self.shape3d) if i != ignore)
        for label, attrs in instance_rles.items():
            box = to_box3d(index2d, attrs['box'], self.axis)
            if self.axis == 'xy':
                starts = attrs['starts'] + index2d * math.prod(shape2d)
                runs = attrs['runs']
            elif self.axis == 'xz':
                hc, wc = np.unravel_index(attrs['starts'], shape2d)
                dc = np.repeat([index2d], len(hc))
                starts = np.ravel_multi_index((hc, dc, wc), self.shape3d)
                runs = attrs['runs']
            else:
                flat = rle_decode(attrs['starts'], attrs['runs'])
                hc, wc = np.unravel_index(flat, shape2d)
                dc = np.repeat([index2d], len(hc))
                starts = np.ravel_multi_index((hc, wc, dc), self.shape3d)
                runs = np.ones_like(starts)
            if label not in self.instances:
                self.instances[label] = {'box': box, 'starts': [starts], 'runs': [runs]}
            else:
                inst = self.instances[label]
                inst['box'] = merge_boxes(box, inst['box'])
                inst['starts'].append(starts)
                inst['runs'].append(runs)

    def finish(self):
        """tracker.py:102-123 -- concat in update order; yz: sort + re-encode."""
        for inst in self.instances.values():
            if isinstance(inst['starts'], list):
                starts = np.concatenate(inst['starts'])
                if self.axis == 'yz':
                    starts, runs = rle_encode(np.sort(starts, kind='stable'))
                else:
                    runs = np.concatenate(inst['runs'])
                inst['starts'] = starts
                inst['runs'] = runs
        self.finished = True


def create_axis_trackers(axes, class_labels, label_divisor, shape):
    """patterns.py:41-53"""
    return {name: [InstanceTracker(c, label_divisor, shape, name) for c in class_labels]
            for name in axes}


def update_trackers(rle_seg, index, trackers):
    """patterns.py:123-134"""
    for t in trackers:
        t.update(rle_seg[t.class_id], index)


def finish_tracking(trackers):
    """patterns.py:136-139"""
    for t in trackers:
        t.finish()


# ----------------------------------------------------------------------------- filters.py
def remove_small_objects(tracker, min_size=64):
    """filters.py:9-24"""
    for k in list(tracker.instances.keys()):
        if tracker.instances[k]['runs'].sum() < min_size:
            del tracker.instances[k]


def remove_pancakes(tracker, min_span=4):
    """filters.py:26-43"""
    for k in list(tracker.instances.keys()):
        b = tracker.instances[k]['box']
        if any(s < min_span for s in (b[3] - b[0], b[4] - b[1], b[5] - b[2])):
            del tracker.instances[k]
