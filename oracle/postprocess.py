"""CPU oracle for the per-slice panoptic post-processing (TEST INFRASTRUCTURE ONLY).

numpy restatement of empanada/inference/postprocess.py and the engine glue of
empanada/inference/engines.py.  All tensors are numpy arrays; fp32 stays fp32.
"""
import ctypes
from collections import deque

import numpy as np

from ._clib import lib

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def factor_pad(x, factor=16):
    """postprocess.py:25-36 -- zero-pad bottom/right of (..., H, W) to a multiple of factor."""
    h, w = x.shape[-2:]
    pb = factor - h % factor if h % factor != 0 else 0
    pr = factor - w % factor if w % factor != 0 else 0
    if pb == 0 and pr == 0:
        return x
    pad = [(0, 0)] * (x.ndim - 2) + [(0, pb), (0, pr)]
    return np.pad(x, pad)


def logits_to_prob(logits):
    """engines.py:22-30 -- softmax over channels if C>1 else sigmoid (fp32).

    Only used on the oracle side to build inputs; parity of the HIP path is
    defined from probabilities onward ("float boundary").
    """
    logits = _f32(logits)
    if logits.shape[1] > 1:
        m = logits.max(axis=1, keepdims=True)
        e = np.exp(logits - m)
        return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    return (1.0 / (1.0 + np.exp(-logits))).astype(np.float32)


def find_instance_center(ctr_hmp, threshold=0.1, nms_kernel=7):
    """postprocess.py:38-76 -> (K,2) int64 centres (y,x) in raster order."""
    hmp = _f32(np.squeeze(ctr_hmp))
    assert hmp.ndim == 2, 'Something is wrong with center heatmap dimension.'
    h, w = hmp.shape
    cap = h * w
    out = np.empty((cap, 2), dtype=np.int64)
    n = lib().emp_oracle_find_centers(
        hmp.ctypes.data_as(_f32p), h, w, ctypes.c_float(threshold), int(nms_kernel),
        out.ctypes.data_as(_i64p), cap)
    return out[:n].copy()


def group_pixels(ctr, offsets, chunksize=20, step=1):
    """postprocess.py:118-169 (+ chunked_pixel_grouping :78-116) -> (1,h,w) int64 ids in 1..K."""
    assert chunksize == 20, "the reference never overrides chunksize"
    ctr = np.ascontiguousarray(ctr, dtype=np.int64)
    assert ctr.shape[0] > 0
    offsets = _f32(offsets)
    if offsets.shape[0] != 1:
        raise ValueError('Only supports inference for batch size = 1')
    offsets = offsets[0]
    _, h, w = offsets.shape
    out = np.empty((h, w), dtype=np.int64)
    lib().emp_oracle_group_pixels(
        ctr.ctypes.data_as(_i64p), ctr.shape[0], offsets.ctypes.data_as(_f32p), h, w,
        int(step), out.ctypes.data_as(_i64p))
    return out[None]


def merge_semantic_and_instance(sem_seg, ins_seg, label_divisor, thing_list, stuff_area, void_label):
    """postprocess.py:223-296 -- majority-class vote per instance, per-class renumbering, stuff area."""
    sem_seg = np.asarray(sem_seg, dtype=np.int64)
    ins_seg = np.asarray(ins_seg, dtype=np.int64)
    pan = np.zeros_like(sem_seg) + void_label
    thing_seg = ins_seg > 0
    sem_thing = np.isin(sem_seg, list(thing_list))

    class_id_tracker = {}
    for ins_id in np.unique(ins_seg):
        if ins_id == 0:
            continue
        mask = (ins_seg == ins_id) & sem_thing
        if not mask.any():
            continue
        # torch.mode: most frequent value, ties -> smallest value
        vals, counts = np.unique(sem_seg[mask], return_counts=True)
        class_id = int(vals[np.argmax(counts)])
        new_id = class_id_tracker.get(class_id, 1)
        class_id_tracker[class_id] = new_id + 1
        pan[mask] = class_id * label_divisor + new_id

    for class_id in np.unique(sem_seg):
        if int(class_id) in thing_list:
            continue
        mask = (sem_seg == class_id) & (~thing_seg)
        if np.count_nonzero(mask) >= stuff_area:
            pan[mask] = class_id * label_divisor
    return pan


def harden_seg(sem, confidence_thr):
    """engines.py:114-121 / patterns.py:242-251 -- (1,C,H,W) fp32 -> (1,1,H,W) int64."""
    sem = _f32(sem)
    if sem.shape[1] > 1:
        return np.argmax(sem, axis=1)[:, None].astype(np.int64)
    return (sem >= np.float32(confidence_thr)).astype(np.int64)


def get_instance_cells(ctr_hmp, offsets, nms_threshold, nms_kernel, coarse_boundaries, upsampling=1):
    """engines.py:257-275 -- centres + grouping (+ zeros if none) + nearest upsample -> (1,1,H,W) fp32."""
    ctr = find_instance_center(ctr_hmp, nms_threshold, nms_kernel)
    step = 4 if coarse_boundaries else 1
    if ctr.shape[0] == 0:
        cells = np.zeros(np.asarray(ctr_hmp).shape, dtype=np.float32)
    else:
        cells = group_pixels(ctr, offsets, step=step).astype(np.float32)[None]
    f = int(upsampling * step)
    if f != 1:
        cells = np.repeat(np.repeat(cells, f, axis=-2), f, axis=-1)  # nearest: out[i] = in[i // f]
    return cells


def get_panoptic_seg(sem, instance_cells, label_divisor, thing_list, stuff_area, void_label):
    """engines.py:277-292 (sem (1,H,W) int64, cells (1,1,H,W) fp32) -> (1,H,W) int64."""
    inst = np.isin(sem, list(thing_list)).astype(np.int64)
    inst = (inst * instance_cells[0]).astype(np.int64)
    return merge_semantic_and_instance(sem, inst, label_divisor, thing_list, stuff_area, void_label)


def get_panoptic_segmentation(sem, ctr_hmp, offsets, thing_list, label_divisor, stuff_area,
                              void_label, threshold=0.1, nms_kernel=7):
    """postprocess.py:298-356 (full-res heads) -> (pan (1,1,H,W), centres (1,K,2))."""
    sem = np.asarray(sem, dtype=np.int64)
    if sem.shape[1] != 1:
        raise ValueError('Expect single channel semantic segmentation. Softmax/argmax first!')
    for t in (sem, ctr_hmp, offsets):
        if np.asarray(t).shape[0] != 1:
            raise ValueError('Only supports inference for batch size = 1')
    sem0 = sem[0]
    ctr = find_instance_center(ctr_hmp, threshold, nms_kernel)
    if ctr.shape[0] == 0:
        ins = np.zeros_like(sem0)
    else:
        ins = np.isin(sem0, list(thing_list)).astype(np.int64) * group_pixels(ctr, offsets)
    # :352-354 passes the un-squeezed (1,1,H,W) sem, so broadcasting makes pan (1,1,H,W)
    pan = merge_semantic_and_instance(sem0, ins, label_divisor, thing_list, stuff_area, void_label)
    return pan[None], ctr[None]


class MedianQueue:
    """engines.py:47-90 -- deque(maxlen=ks); the median is written back into the middle item."""

    def __init__(self, median_kernel_size):
        assert median_kernel_size % 2 == 1, "Kernel size must be odd integer!"
        self.ks = median_kernel_size
        self.mid_idx = (median_kernel_size - 1) // 2
        self.median_queue = deque(maxlen=median_kernel_size)

    def reset(self):
        self.median_queue = deque(maxlen=self.ks)

    def get_median(self, key):
        stack = np.concatenate([_f32(o[key]) for o in self.median_queue], axis=0)
        # torch.median over dim 0 of an odd count == the middle order statistic
        return np.sort(stack, axis=0)[self.mid_idx][None]

    def enqueue(self, item):
        self.median_queue.append(item)

    def get_next(self, keys):
        nq = len(self.median_queue)
        if nq <= self.mid_idx:
            return self.median_queue[-1]
        if nq < self.ks:
            return None
        out = self.median_queue[self.mid_idx]
        for key in keys:
            out[key] = self.get_median(key)
        return out

    def end(self):
        return list(self.median_queue)[self.mid_idx + 1:]


def post_slice(o, *, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1, nms_kernel=7,
               confidence_thr=0.5, coarse_boundaries=True, render=True):
    """What a 3d engine does with one item leaving the median queue: PanopticDeepLabRenderEngine3d.postprocess
    (engines.py:344-349 -> :277-292, cropped to the item's size) or PanopticDeepLabEngine3d (engines.py:209-219)."""
    if render:
        cells = get_instance_cells(o['ctr_hmp'], o['offsets'], nms_threshold, nms_kernel, coarse_boundaries, 1)
        sem = harden_seg(o['sem'], confidence_thr)[0]
        pan = get_panoptic_seg(sem, cells, label_divisor, thing_list, stuff_area, void_label)
        h, w = o['size']
        return pan[..., :h, :w]
    sem = harden_seg(o['sem'], confidence_thr)
    pan, _ = get_panoptic_segmentation(sem, o['ctr_hmp'], o['offsets'], thing_list, label_divisor,
                                       stuff_area, void_label, nms_threshold, nms_kernel)
    return pan


def engine3d_stack(sem_probs, ctr_hmps, offsets, *, thing_list, label_divisor=1000, stuff_area=64,
                   void_label=0, nms_threshold=0.1, nms_kernel=7, confidence_thr=0.5,
                   median_kernel_size=3, coarse_boundaries=True, render=True, sizes=None):
    """Drive the reference's 3d engine call sequence over a stack of head tensors.

    render=True : PanopticDeepLabRenderEngine3d.__call__/end  (engines.py:351-394)
    render=False: PanopticDeepLabEngine3d.__call__/end        (engines.py:183-221)
    sem_probs[t] (1,C,Hp,Wp), ctr_hmps[t] (1,1,h,w), offsets[t] (1,2,h,w).
    Returns the list of emitted pan segs in emission order (Nones dropped) --
    i.e. what scripts/pdl_inference3d.py:163-182 puts on the matcher queue.
    """
    q = MedianQueue(median_kernel_size)
    outs = []
    kw = dict(thing_list=thing_list, label_divisor=label_divisor, stuff_area=stuff_area, void_label=void_label,
              nms_threshold=nms_threshold, nms_kernel=nms_kernel, confidence_thr=confidence_thr,
              coarse_boundaries=coarse_boundaries, render=render)

    for t in range(len(sem_probs)):
        size = sizes[t] if sizes is not None else tuple(np.asarray(sem_probs[t]).shape[-2:])
        q.enqueue({'sem': _f32(sem_probs[t]), 'ctr_hmp': _f32(ctr_hmps[t]),
                   'offsets': _f32(offsets[t]), 'size': size})
        o = q.get_next(['sem'])
        if o is not None:
            outs.append(post_slice(o, **kw))
    for o in q.end():
        outs.append(post_slice(o, **kw))
    return outs
