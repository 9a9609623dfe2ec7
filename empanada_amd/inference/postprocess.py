"""Panoptic-DeepLab post-processing with the reference's names and signatures
(``empanada/inference/postprocess.py``), torch tensors in / torch tensors out, computed by the
HIP kernels of libemp_hip.so.  Inputs may live on the host; they are moved to the GPU.

  factor_pad                     :25-36    (F.pad, memory plumbing)
  find_instance_center           :38-76    -> emp_find_centers
  group_pixels                   :118-169  -> emp_group_pixels (incl. chunked_pixel_grouping :78-116)
  get_instance_segmentation      :171-221
  merge_semantic_and_instance    :223-296  -> emp_fuse_panoptic
  get_panoptic_segmentation      :298-356
"""
from typing import List

import torch
import torch.nn.functional as F

from .. import _hip

__all__ = ['factor_pad', 'find_instance_center', 'group_pixels', 'get_instance_segmentation',
           'merge_semantic_and_instance', 'get_panoptic_segmentation']

_CAPS = (256, 1024, _hip.MAX_CENTERS)


def _cuda(t):
    _hip.require_gpu()
    return t if t.is_cuda else t.cuda()


def factor_pad(tensor, factor: int = 16):
    """postprocess.py:25-36"""
    h, w = tensor.size()[2:]
    pad_bottom = factor - h % factor if h % factor != 0 else 0
    pad_right = factor - w % factor if w % factor != 0 else 0
    if pad_bottom == 0 and pad_right == 0:
        return tensor
    return F.pad(tensor, (0, pad_right, 0, pad_bottom))


def centers_batched(ctr_hmp, threshold, nms_kernel):
    """(D,1,h,w) or (D,h,w) fp32 -> (idx (D,cap) int32, count (D) int32) on the device, raster order.
    Grows the capacity when a slice overflows; raises beyond EMP_MAX_CENTERS."""
    hm = _cuda(ctr_hmp).float()
    if hm.dim() == 4:
        hm = hm[:, 0]
    hm = hm.contiguous()
    for cap in _CAPS:
        idx, cnt = _hip.find_centers(hm, threshold, nms_kernel, cap=cap)
        if int(cnt.max().item()) <= cap:
            return idx, cnt
    raise _hip.HipError(f"more than {_hip.MAX_CENTERS} instance centres in one slice")


def find_instance_center(ctr_hmp, threshold: float = 0.1, nms_kernel: int = 7):
    """postprocess.py:38-76 -> (K, 2) int64 (y, x) in raster order."""
    hm = ctr_hmp.squeeze()
    assert len(hm.size()) == 2, 'Something is wrong with center heatmap dimension.'
    idx, cnt = centers_batched(hm[None], threshold, nms_kernel)
    k = int(cnt[0].item())
    flat = idx[0, :k].long()
    w = hm.size(1)
    return torch.stack([flat // w, flat % w], dim=1)


def _ctr_to_idx(ctr, w):
    ctr = _cuda(ctr).long()
    K = ctr.size(0)
    if K > _hip.MAX_CENTERS:
        raise _hip.HipError(f"more than {_hip.MAX_CENTERS} instance centres in one slice")
    idx = (ctr[:, 0] * w + ctr[:, 1]).int().reshape(1, K).contiguous()
    cnt = torch.full((1,), K, dtype=torch.int32, device=idx.device)
    return idx, cnt


def group_pixels(ctr, offsets, chunksize: int = 20, step: float = 1):
    """postprocess.py:118-169 -> (1, h, w) int64 ids in 1..K (0 where every centre is >= 1e5 away, K > 20)."""
    assert ctr.size(0) > 0
    if offsets.size(0) != 1:
        raise ValueError('Only supports inference for batch size = 1')
    assert chunksize == 20, "the reference never overrides chunksize; the 20-centre chunk rule is built in"
    offsets = _cuda(offsets).float().contiguous()
    idx, cnt = _ctr_to_idx(ctr, offsets.size(3))
    ids = _hip.group_pixels(idx, cnt, offsets, int(step))
    return ids.view(torch.int16).long() & 0xFFFF


def merge_semantic_and_instance(sem_seg, ins_seg, label_divisor: int, thing_list: List[int], stuff_area: int,
                                void_label: int):
    """postprocess.py:223-296.  sem_seg, ins_seg: integer tensors of equal (or broadcastable) shape whose last two
    dims are (H, W) and whose leading dims multiply to 1.  Returns int64 of the broadcast shape."""
    sem_seg, ins_seg = _cuda(sem_seg), _cuda(ins_seg)
    shape = torch.broadcast_shapes(sem_seg.shape, ins_seg.shape)
    H, W = shape[-2:]
    sem = sem_seg.reshape(1, H, W)
    ins = ins_seg.reshape(1, H, W)
    if int(sem.max().item()) >= _hip.MAX_CLASSES or int(sem.min().item()) < 0:
        raise ValueError(f"class ids must be in [0, {_hip.MAX_CLASSES})")
    # instance ids are arbitrary integers: rank them (ascending, 0 stays 0) for the kernel
    uniq, inv = torch.unique(ins, return_inverse=True)
    rank = inv if int(uniq[0].item()) == 0 else inv + 1
    cap = max(int(rank.max().item()), 1)
    if cap > 65535:
        raise ValueError("more than 65535 distinct instance ids")
    n_classes = max(int(sem.max().item()) + 1, max(thing_list) + 1 if len(thing_list) else 1)
    n_classes = min(n_classes, _hip.MAX_CLASSES)
    things = [t for t in thing_list if t < n_classes]
    pan = _hip.fuse_panoptic(sem.to(torch.uint8).contiguous(), rank.to(torch.int16).contiguous().view(torch.uint16),
                             cap, n_classes, things, label_divisor, stuff_area, void_label, up=1,
                             out_dtype=torch.int64)
    return pan.reshape(shape)


def get_instance_segmentation(sem_seg, ctr_hmp, offsets, thing_list: List[int], threshold: float = 0.1,
                              nms_kernel: int = 7):
    """postprocess.py:171-221 -> (thing_seg (1,H,W) int64, centres (1,K,2))."""
    assert sem_seg.size(0) == 1, 'Only batch size of 1 is supported!'
    sem_seg = _cuda(sem_seg)[0]
    instance_seg = torch.zeros_like(sem_seg)
    for thing_class in thing_list:
        instance_seg[sem_seg == thing_class] = 1
    ctr = find_instance_center(ctr_hmp, threshold=threshold, nms_kernel=nms_kernel)
    if ctr.size(0) == 0:
        return torch.zeros_like(sem_seg), ctr.unsqueeze(0)
    instance_id = group_pixels(ctr, offsets)
    return instance_seg * instance_id, ctr.unsqueeze(0)


def get_panoptic_segmentation(sem, ctr_hmp, offsets, thing_list: List[int], label_divisor: int, stuff_area: int,
                              void_label: int, threshold: float = 0.1, nms_kernel: int = 7):
    """postprocess.py:298-356 -> (pan (1,1,H,W) int64, centres (1,K,2))."""
    if sem.size(1) != 1:
        raise ValueError('Expect single channel semantic segmentation. Softmax/argmax first!')
    if sem.size(0) != 1:
        raise ValueError('Only supports inference for batch size = 1')
    if ctr_hmp.size(0) != 1:
        raise ValueError('Only supports inference for batch size = 1')
    if offsets.size(0) != 1:
        raise ValueError('Only supports inference for batch size = 1')
    instance, center = get_instance_segmentation(sem, ctr_hmp, offsets, thing_list, threshold=threshold,
                                                 nms_kernel=nms_kernel)
    panoptic = merge_semantic_and_instance(_cuda(sem), instance, label_divisor, thing_list, stuff_area, void_label)
    return panoptic, center


# ----------------------------------------------------------------------------- batched fast path
def panoptic_stack(sem_prob, ctr_hmp, offsets, *, thing_list, label_divisor=1000, stuff_area=64, void_label=0,
                   nms_threshold=0.1, nms_kernel=7, confidence_thr=0.5, median_kernel_size=3, coarse_boundaries=True,
                   upsampling=1, n_classes=None, out_dtype=torch.uint32):
    """Whole-stack form of the 3d engines (engines.py:161-221, 327-394): everything from probabilities
    to panoptic labels for D slices in five kernel groups, no host round trip per slice.

    sem_prob (D,C,Hp,Wp) fp32, ctr_hmp (D,1,h,w), offsets (D,2,h,w) with h = Hp/step, step = 4 if
    coarse_boundaries else 1.  Returns (pan (D',Hp,Wp) uint32 device tensor, emitted slice indices):
    D' == D unless the stack is shorter than the median kernel, in which case the slices the reference's
    queue loses (engines.py:68-90) are dropped here as well.
    """
    _hip.require_gpu()
    sem_prob = _cuda(sem_prob).float().contiguous()
    D, C, Hp, Wp = sem_prob.shape
    ks = int(median_kernel_size)
    m = (ks - 1) // 2
    emitted = list(range(D))
    if D < ks:
        emitted = list(range(min(D, m))) + list(range(m + 1, D))
        sel = torch.tensor(emitted, dtype=torch.long, device=sem_prob.device)
        sem_prob = sem_prob[sel].contiguous()
        ctr_hmp = _cuda(ctr_hmp)[sel]
        offsets = _cuda(offsets)[sel]
        ks = 1
        D = len(emitted)
    if D == 0:
        empty = torch.zeros((0, Hp, Wp), dtype=torch.int32 if out_dtype == torch.uint32 else out_dtype,
                            device=sem_prob.device)
        return (empty.view(torch.uint32) if out_dtype == torch.uint32 else empty), emitted
    sem = _hip.median_harden_stack(sem_prob, ks, confidence_thr)
    step = 4 if coarse_boundaries else 1
    idx, cnt = centers_batched(ctr_hmp, nms_threshold, nms_kernel)
    # full-resolution heads: only thing pixels are voted on (the fusion masks the rest anyway)
    ids = _hip.group_pixels(idx, cnt, _cuda(offsets).float().contiguous(), step,
                            sem=sem if (step == 1 and upsampling == 1) else None, thing_list=thing_list)
    if n_classes is None:
        n_classes = max(2 if C == 1 else C, max(thing_list) + 1)
    pan = _hip.fuse_panoptic(sem, ids, idx.shape[1], n_classes, thing_list, label_divisor, stuff_area, void_label,
                             up=int(step * upsampling), out_dtype=out_dtype)
    return pan, emitted
