"""RLE / box / range primitives with the reference's names and signatures
(``empanada/array_utils.py``), numpy in / numpy out, heavy lifting in libemp_hip.so.

  rle_intersection / rle_iou / rle_ioa   -> emp_rle_pair_intersections  (array_utils.py:340-455)
  vote_by_ranges / join_ranges / merge_rles / rle_voting -> emp_vote_ranges (:457-723)
  numpy_fill_instances                   -> emp_fill_runs_u32 / _u8      (:725-737)
Pure bookkeeping (boxes, encode/decode of index lists, strings, take/put) stays on the host,
as in the reference.  There is no CPU implementation of the kernels above in this package.
"""
import math

import numpy as np
import torch

from . import _hip

__all__ = [
    'take', 'put', 'box_area', 'merge_boxes', 'box_iou', 'rle_encode', 'rle_decode', 'rle_to_string',
    'string_to_rle', 'rle_intersection', 'rle_iou', 'rle_ioa', 'rle_voting', 'vote_by_ranges', 'rle_to_ranges',
    'ranges_to_rle', 'concat_sort_ranges', 'join_ranges', 'merge_rles', 'numpy_fill_instances',
    'rle_pair_intersections',
]


def _dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


# ----------------------------------------------------------------------------- host bookkeeping
def take(array, indices, axis=0):
    """array_utils.py:6-23"""
    return array[tuple(slice(None) if n != axis else indices for n in range(array.ndim))]


def put(array, indices, value, axis=0):
    """array_utils.py:25-40"""
    array[tuple(slice(None) if n != axis else indices for n in range(array.ndim))] = value


def box_area(boxes):
    """array_utils.py:42-59"""
    boxes = np.asarray(boxes)
    nd = boxes.shape[1] // 2
    return math.prod(boxes[:, i + nd] - boxes[:, i] for i in range(nd))


def merge_boxes(box1, box2):
    """array_utils.py:101-125"""
    n = len(box1)
    nd = n // 2
    return tuple(min(box1[i], box2[i]) if i < nd else max(box1[i], box2[i]) for i in range(n))


def box_pairs(boxes1, boxes2=None):
    """Pairs of boxes with strictly positive intersection (array_utils.py:144-172), row-major.
    Returns rows, cols, ious (fp64), intersections (int64).  Which pairs intersect is decided on the GPU
    (emp_box_pairs); the IoU of the surviving pairs is integer arithmetic on the host."""
    boxes1 = np.asarray(boxes1, dtype=np.int64)
    boxes2 = boxes1 if boxes2 is None else np.asarray(boxes2, dtype=np.int64)
    if boxes1.size == 0 or boxes2.size == 0:
        e = np.zeros(0, dtype=np.int64)
        return e, e, np.zeros(0), e
    _hip.require_gpu()
    if max(int(np.abs(boxes1).max()), int(np.abs(boxes2).max())) >= 2 ** 31:
        raise ValueError("box coordinates must fit in int32")
    pairs = _hip.box_pairs(_dev(boxes1, np.int32), _dev(boxes2, np.int32)).cpu().numpy().astype(np.int64)
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))                  # row-major like the reference's nested loops
    rows, cols = pairs[order, 0], pairs[order, 1]
    nd = boxes1.shape[1] // 2
    b1, b2 = boxes1[rows], boxes2[cols]
    it = np.prod(np.minimum(b1[:, nd:], b2[:, nd:]) - np.maximum(b1[:, :nd], b2[:, :nd]), axis=1)
    a1 = np.prod(b1[:, nd:] - b1[:, :nd], axis=1)
    a2 = np.prod(b2[:, nd:] - b2[:, :nd], axis=1)
    return rows, cols, it / (a1 + a2 - it), it


def box_iou(boxes1, boxes2=None, return_intersection=False):
    """array_utils.py:174-207 -> scipy CSR matrices like the reference."""
    from scipy.sparse import csr_matrix
    b2 = boxes1 if boxes2 is None else boxes2
    shape = (len(boxes1), len(b2))
    rows, cols, ious, inter = box_pairs(boxes1, boxes2)
    iou_csr = csr_matrix((ious, (rows, cols)), shape=shape)
    if return_intersection:
        return iou_csr, csr_matrix((inter, (rows, cols)), shape=shape)
    return iou_csr


def rle_encode(indices):
    """array_utils.py:209-235 (emp_rle_encode)"""
    indices = np.asarray(indices)
    if len(indices) == 0:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")     # what the reference does
    _hip.require_gpu()
    st, rn = _hip.rle_encode(_dev(indices, np.int64))
    return st.cpu().numpy().astype(indices.dtype, copy=False), rn.cpu().numpy()


def rle_decode(starts, runs):
    """array_utils.py:237-252 (emp_rle_decode)"""
    if len(starts) == 0:
        raise ValueError("need at least one array to concatenate")
    _hip.require_gpu()
    return _hip.rle_decode(_dev(starts, np.int64), _dev(runs, np.int64)).cpu().numpy()


def rle_to_string(starts, runs):
    """array_utils.py:254-267"""
    return ' '.join(f'{i} {r}' for i, r in zip(starts, runs))


def string_to_rle(encoding):
    """array_utils.py:269-283"""
    enc = np.array([int(i) for i in encoding.split(' ')])
    return enc[::2], enc[1::2]


def rle_to_ranges(rle):
    """array_utils.py:617-618"""
    return np.cumsum(rle, axis=1)


def ranges_to_rle(ranges):
    """array_utils.py:620-623"""
    ranges = ranges.copy()
    ranges[:, 1] = ranges[:, 1] - ranges[:, 0]
    return ranges


def concat_sort_ranges(list_of_ranges):
    """array_utils.py:625-632"""
    lst = [np.asarray(r) for r in list_of_ranges if len(r) > 0]
    ranges = np.concatenate(lst, axis=0)
    return ranges[np.argsort(ranges[:, 0], kind='stable')]


# ----------------------------------------------------------------------------- HIP-backed

def rle_pair_intersections(starts_list, runs_list, pairs):
    """Intersections for many (a, b) pairs of instances at once (emp_rle_pair_intersections).

    starts_list / runs_list: per-instance int64 arrays; pairs (n,2) int.  Returns int64 (n,).
    Each instance's runs are put in stable start order first (the kernel merges two sorted
    lists, A before B on ties -- equal to the reference's stable argsort of the concatenation).
    """
    _hip.require_gpu()
    pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    if len(pairs) == 0:
        return np.zeros(0, dtype=np.int64)
    sizes = np.array([len(s) for s in starts_list], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    st = np.concatenate([np.asarray(s, dtype=np.int64) for s in starts_list]) if off[-1] else np.zeros(0, np.int64)
    ln = np.concatenate([np.asarray(r, dtype=np.int64) for r in runs_list]) if off[-1] else np.zeros(0, np.int64)
    inst = np.repeat(np.arange(len(sizes)), sizes)
    order = np.lexsort((np.arange(len(st)), st, inst))      # by instance, then start, stable
    out = _hip.rle_pair_intersections(_dev(st[order], np.int64), _dev(ln[order], np.int64), _dev(off, np.int64),
                                      _dev(pairs, np.int32))
    return out.cpu().numpy()


def rle_intersection(starts_a, runs_a, starts_b, runs_b):
    """array_utils.py:371-403"""
    return int(rle_pair_intersections([starts_a, starts_b], [runs_a, runs_b], [[0, 1]])[0])


def rle_iou(starts_a, runs_a, starts_b, runs_b, return_intersection=False):
    """array_utils.py:405-429"""
    inter = rle_intersection(starts_a, runs_a, starts_b, runs_b)
    union = int(np.sum(runs_a)) + int(np.sum(runs_b)) - inter
    iou = np.float64(inter) / np.float64(union)
    return (iou, inter) if return_intersection else iou


def rle_ioa(starts_a, runs_a, starts_b, runs_b, return_intersection=False):
    """array_utils.py:431-455"""
    inter = rle_intersection(starts_a, runs_a, starts_b, runs_b)
    ioa = np.float64(inter) / np.float64(int(np.sum(runs_b)))
    return (ioa, inter) if return_intersection else ioa


def vote_groups(list_of_groups, vote_thr):
    """Coverage voting for many groups at once: list_of_groups[g] is a list of (n_i,2) range arrays.
    Returns a list of (m_g,2) int64 arrays (emp_vote_ranges)."""
    _hip.require_gpu()
    st, en, gr = [], [], []
    for g, lst in enumerate(list_of_groups):
        for r in lst:
            r = np.asarray(r, dtype=np.int64).reshape(-1, 2)
            st.append(r[:, 0]); en.append(r[:, 1]); gr.append(np.full(len(r), g, dtype=np.int32))
    ng = len(list_of_groups)
    if not st or sum(len(s) for s in st) == 0:
        return [np.zeros((0, 2), dtype=np.int64) for _ in range(ng)]
    out, off = _hip.vote_ranges(_dev(np.concatenate(st), np.int64), _dev(np.concatenate(en), np.int64),
                                _dev(np.concatenate(gr), np.int32), ng, int(vote_thr))
    off = off.cpu().numpy()
    out = out[:int(off[-1])].cpu().numpy()
    return [out[off[g]:off[g + 1]] for g in range(ng)]


def rle_voting(ranges, vote_thr=2, init_index=None, term_index=None):
    """array_utils.py:539-601 (init_index / term_index are never used by the reference's callers)."""
    assert vote_thr > 1, "For vote_thr of 1 use join_ranges instead!"
    assert init_index is None and term_index is None, "init_index/term_index are not supported"
    ranges = np.asarray(ranges)
    if len(ranges) < 2:
        return []          # the reference's pairwise loop never runs
    return vote_groups([[ranges]], vote_thr)[0].tolist()


def join_ranges(list_of_ranges):
    """array_utils.py:665-671 (+ _join_ranges :634-663, incl. its single-range UnboundLocalError)."""
    lst = [np.asarray(r) for r in list_of_ranges if len(r) > 0]
    if sum(len(r) for r in lst) < 2:
        if not lst:
            raise ValueError("need at least one array to concatenate")      # np.concatenate([]) in the reference
        raise UnboundLocalError("local variable 'range2' referenced before assignment")
    return vote_groups([lst], 1)[0]


def vote_by_ranges(list_of_ranges, vote_thr=2):
    """array_utils.py:603-615"""
    lst = [r for r in list_of_ranges if len(r) > 0]
    if vote_thr == 1:
        return join_ranges(lst)
    if len(lst) >= vote_thr:
        n = sum(len(r) for r in lst)
        if n < 2:
            return np.array([])
        out = vote_groups([lst], vote_thr)[0]
        return out if len(out) else np.array([])
    return np.array([])


def merge_rles(starts_a, runs_a, starts_b=None, runs_b=None):
    """array_utils.py:690-723"""
    lst = [np.stack([starts_a, starts_a + runs_a], axis=1)]
    if starts_b is not None and runs_b is not None:
        lst.append(np.stack([starts_b, starts_b + runs_b], axis=1))
    joined = ranges_to_rle(join_ranges(lst))
    return joined[:, 0], joined[:, 1]


def numpy_fill_instances(volume, instances):
    """array_utils.py:725-737: paint the instances into `volume` (numpy, in place) on the GPU.  Instance ids are labels
    (class * divisor + n >= 1); an id of 0 paints nothing (emp_fill_runs_u32 uses 0 for "skip this instance")."""
    _hip.require_gpu()
    ids = list(instances.keys())
    if not ids:
        return volume
    starts = [np.asarray(instances[k]['starts'], dtype=np.int64) for k in ids]
    runs = [np.asarray(instances[k]['runs'], dtype=np.int64) for k in ids]
    order = np.repeat(np.arange(len(ids), dtype=np.int32), [len(s) for s in starts])
    flat = volume.reshape(-1)
    top = int(flat.max()) if flat.size else 0
    if volume.dtype.itemsize > 4 or int(max(ids)) >= 2 ** 31 or top >= 2 ** 31:
        raise ValueError("fill: ids must be < 2^31 and the volume at most 32-bit")
    if top > 0 or (volume.dtype.kind == 'i' and flat.size and int(flat.min()) < 0):
        dvol = _hip.np_to_dev_u32(flat)
    else:                                        # a fresh volume (the usual case): nothing to upload
        dvol = torch.zeros((flat.size,), dtype=torch.int32, device='cuda').view(torch.uint32)
    _hip.fill_runs_u32(dvol, _dev(np.concatenate(starts), np.int64), _dev(np.concatenate(runs), np.int64),
                       _dev(order, np.int32), _hip.np_to_dev_u32(np.asarray(ids, dtype=np.int64)))
    if volume.dtype.itemsize == 4 and volume.dtype.kind in 'ui' and flat.flags.c_contiguous and flat.flags.writeable \
            and np.shares_memory(flat, volume):
        torch.from_numpy(flat.view(np.int32)).copy_(dvol.view(torch.int32))     # straight into the caller's memory
    else:
        flat[:] = dvol.cpu().numpy().astype(volume.dtype)
    return flat.reshape(volume.shape)
