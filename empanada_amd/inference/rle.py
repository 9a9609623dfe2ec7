"""pan_seg <-> run-length-encoded segmentation, HIP-backed.

Mirror of the reference's ``empanada/inference/rle.py`` (same names, argument meaning and
returned structures): ``pan_seg_to_rle_seg`` :26-86, ``rle_seg_to_pan_seg`` :88-118,
``unpack_rle_attrs`` :120-150, ``connected_components`` :18-24.

The pixel work (run extraction, 8-connected components, boxes, areas) runs in libemp_hip.so
(emp_runs_count / emp_runs_extract / emp_runs_label); the host only regroups the O(#runs) run
table into the reference's dict-of-dicts.
"""
import numpy as np
import torch

from .. import _hip
from ..array_utils import string_to_rle
from .deferred import LazyPan

__all__ = ['pan_seg_to_rle_seg', 'rle_seg_to_pan_seg', 'unpack_rle_attrs', 'connected_components',
           'stack_to_rle_segs', 'runs_to_instances']


def _to_device_u32(pan_seg):
    _hip.require_gpu()
    if isinstance(pan_seg, torch.Tensor):
        return _hip.as_u32(pan_seg.to('cuda'))
    a = np.asarray(pan_seg)
    if a.size and (a.min() < 0 or a.max() >= 2 ** 32):
        raise ValueError("panoptic labels must fit in uint32")
    return _hip.np_to_dev_u32(a)


def runs_to_instances(r_start, r_len, r_comp, n_comp):
    """Group a raster-ordered run table by component and merge runs that are contiguous in flat
    index (a run ending at the last column continues at column 0 of the next row -- exactly what
    rle_encode of the flat indices yields, array_utils.py:209-235).

    Returns (starts, runs, off): component c owns [off[c], off[c+1]) of the int64 arrays.
    """
    if len(r_start) == 0:
        return (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(n_comp + 1, np.int64))
    order = np.argsort(r_comp, kind='stable')
    comp = r_comp[order]
    st = r_start[order].astype(np.int64)
    ln = r_len[order].astype(np.int64)
    brk = np.ones(len(st), dtype=bool)
    brk[1:] = (comp[1:] != comp[:-1]) | (st[1:] != st[:-1] + ln[:-1])
    seg = np.flatnonzero(brk)
    starts = st[seg]
    runs = np.add.reduceat(ln, seg)
    seg_comp = comp[seg]
    off = np.searchsorted(seg_comp, np.arange(n_comp + 1), side='left').astype(np.int64)
    return starts, runs, off


def stack_to_rle_segs(pan, labels, label_divisor, thing_list, force_connected=True):
    """Batched form: pan (D,H,W) uint32 cuda tensor -> (list of D rle_seg dicts, RunTable).

    One pass of the HIP run/CC kernels over the whole stack; the reference calls
    pan_seg_to_rle_seg once per slice (inference/patterns.py:93-95).
    """
    labels = list(labels)
    cc = [l for l in labels if (force_connected and l in thing_list)]
    table = _hip.extract_runs(pan, label_divisor, cc)
    D = table.D
    # one device-to-host copy for the whole table (eight copies were eight synchronisations per call -- per SLICE in the
    # per-slice protocol): every column as int64 in one buffer, cut up on the host
    nr, nc = int(table.n_runs), int(table.n_comp)
    cols = [table.r_start, table.r_len, table.r_comp, table.r_val.view(torch.int32), table.c_slice, table.c_label,
            table.c_box.reshape(-1), table.c_first]
    flat = torch.cat([c.reshape(-1).to(torch.int64) for c in cols]).cpu().numpy() if nr else np.zeros(0, np.int64)
    cut = np.cumsum([0, nr, nr, nr, nr, nc, nc, 4 * nc, nc])
    part = lambda i: flat[cut[i]:cut[i + 1]] if nr else np.zeros(0, np.int64)
    r_start, r_len, r_comp = part(0).astype(np.int32), part(1).astype(np.int32), part(2).astype(np.int32)
    r_val = (part(3) & 0xFFFFFFFF).astype(np.uint32)                 # the int32 view sign-extends labels >= 2^31
    c_slice, c_label = part(4).astype(np.int32), part(5)
    c_box, c_first = part(6).astype(np.int32).reshape(-1, 4), part(7)
    # the class of a component is that of its ORIGINAL value: relabelled ids may run past the divisor
    c_val = r_val[c_first] if nc else np.zeros(0, np.uint32)
    starts, runs, off = runs_to_instances(r_start, r_len, r_comp, table.n_comp)
    segs = [{l: {} for l in labels} for _ in range(D)]
    if table.n_comp:
        cls = c_val.astype(np.int64) // label_divisor
        # instances of one class come out in ascending label order (regionprops order, rle.py:75)
        for c in np.lexsort((c_label, cls, c_slice)):
            k = int(cls[c])
            if k not in segs[0]:
                continue
            segs[int(c_slice[c])][k][int(c_label[c])] = {
                'box': tuple(int(b) for b in c_box[c]),
                'starts': starts[off[c]:off[c + 1]], 'runs': runs[off[c]:off[c + 1]]}
    return segs, table


def pan_seg_to_rle_seg(pan_seg, labels, label_divisor, thing_list, force_connected=True):
    """Reference signature (rle.py:26-86).  pan_seg: (h, w) array or tensor of panoptic labels -- or the handle a
    deferred 3d engine hands out (inference/deferred.py), in which case a handle comes back."""
    if isinstance(pan_seg, LazyPan):
        seg = pan_seg._s.lazy_rle(pan_seg, (labels, label_divisor, thing_list, force_connected))
        if seg is not None:
            return seg
        pan_seg = pan_seg._force()
    pan = _to_device_u32(pan_seg)
    assert pan.dim() == 2, "pan_seg must be (h, w)"
    segs, _ = stack_to_rle_segs(pan[None].contiguous(), labels, label_divisor, thing_list, force_connected)
    return segs[0]


def connected_components(seg):
    """rle.py:18-24: multi-value 8-connected labelling, ids 1..n in raster order of first pixel."""
    seg_np = np.asarray(seg.cpu() if isinstance(seg, torch.Tensor) else seg)
    pan = _to_device_u32(seg_np)
    # one pseudo class spanning every value -> all runs are split into components numbered 1..n
    div = int(seg_np.max()) + 1 if seg_np.size else 1
    table = _hip.extract_runs(pan[None].contiguous(), div, [0])
    out = torch.zeros(seg_np.size, dtype=torch.int32, device='cuda').view(torch.uint32)
    if table.n_runs:
        ids = _hip.as_u32(table.c_label)
        _hip.fill_runs_u32(out, table.r_start.to(torch.int64), table.r_len.to(torch.int64), table.r_comp, ids)
    return out.reshape(seg_np.shape).cpu().numpy()


def rle_seg_to_pan_seg(rle_seg, shape):
    """rle.py:88-118 -> (h, w) uint32 numpy array (painted on the GPU by emp_fill_runs_u32)."""
    _hip.require_gpu()
    ids, starts, runs, order = [], [], [], []
    for attrs in rle_seg.values():
        for object_id, a in attrs.items():
            order.append(np.full(len(a['starts']), len(ids), dtype=np.int32))
            ids.append(object_id)
            starts.append(np.asarray(a['starts'], dtype=np.int64))
            runs.append(np.asarray(a['runs'], dtype=np.int64))
    n = int(np.prod(shape))
    vol = torch.zeros((n,), dtype=torch.int32, device='cuda').view(torch.uint32)
    if ids and sum(len(s) for s in starts):
        dev = lambda x, dt: torch.from_numpy(np.concatenate(x).astype(dt)).cuda()
        _hip.fill_runs_u32(vol, dev(starts, np.int64), dev(runs, np.int64), dev(order, np.int32),
                           _hip.np_to_dev_u32(np.asarray(ids, dtype=np.int64)))
    return vol.cpu().numpy().reshape(shape)


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
