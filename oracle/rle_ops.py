"""CPU oracle for the RLE / box / range primitives (TEST INFRASTRUCTURE ONLY).

numpy restatement of empanada/array_utils.py.  Index arrays are int64, ranges are
half-open [start, end).
"""
import math

import numpy as np


# ----------------------------------------------------------------------------- boxes
def box_area(boxes):
    """array_utils.py:42-59"""
    boxes = np.asarray(boxes)
    nd = boxes.shape[1] // 2
    return math.prod(boxes[:, i + nd] - boxes[:, i] for i in range(nd))


def merge_boxes(box1, box2):
    """array_utils.py:101-125 -- enclosing box, as a tuple."""
    n = len(box1)
    nd = n // 2
    return tuple(min(box1[i], box2[i]) if i < nd else max(box1[i], box2[i]) for i in range(n))


def box_pairs(boxes1, boxes2=None):
    """array_utils.py:144-172 (_box_iou) -- all pairs with strictly positive intersection.

    Returns rows, cols, ious (fp64), intersects (int64) in row-major (x, y) order.
    The early `break` at :163-164 only skips work: area terms past the break are
    unused because the pair is dropped.
    """
    boxes1 = np.asarray(boxes1, dtype=np.int64)
    boxes2 = boxes1 if boxes2 is None else np.asarray(boxes2, dtype=np.int64)
    if boxes1.size == 0 or boxes2.size == 0:
        e = np.zeros(0, dtype=np.int64)
        return e, e, np.zeros(0), e
    nd = boxes1.shape[1] // 2
    inter = np.ones((len(boxes1), len(boxes2)), dtype=np.int64)
    a1 = np.ones(len(boxes1), dtype=np.int64)
    a2 = np.ones(len(boxes2), dtype=np.int64)
    for i in range(nd):
        lo = np.maximum(boxes1[:, None, i], boxes2[None, :, i])
        hi = np.minimum(boxes1[:, None, i + nd], boxes2[None, :, i + nd])
        inter *= np.maximum(0, hi - lo)
        a1 *= boxes1[:, i + nd] - boxes1[:, i]
        a2 *= boxes2[:, i + nd] - boxes2[:, i]
    rows, cols = np.nonzero(inter > 0)
    it = inter[rows, cols]
    ious = it / (a1[rows] + a2[cols] - it)
    return rows, cols, ious, it


def box_iou_dense(boxes1, boxes2=None):
    """array_utils.py:174-207 as a dense (n,m) fp64 matrix (the reference returns scipy CSR)."""
    boxes1 = np.asarray(boxes1)
    b2 = boxes1 if boxes2 is None else np.asarray(boxes2)
    out = np.zeros((len(boxes1), len(b2)))
    r, c, iou, _ = box_pairs(boxes1, boxes2)
    out[r, c] = iou
    return out


# ----------------------------------------------------------------------------- rle <-> indices
def rle_encode(indices):
    """array_utils.py:209-235"""
    indices = np.asarray(indices)
    changes = np.where(indices[1:] != indices[:-1] + 1)[0] + 1
    changes = np.concatenate([[0], changes, [len(indices)]]).astype(np.int64)
    runs = changes[1:] - changes[:-1]
    return indices[changes[:-1]], runs


def rle_decode(starts, runs):
    """array_utils.py:237-252"""
    return np.concatenate([np.arange(s, s + r) for s, r in zip(starts, runs)])


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


# ----------------------------------------------------------------------------- intersections
def rle_intersection(starts_a, runs_a, starts_b, runs_b):
    """array_utils.py:371-403 + intersection_from_ranges :340-369.

    Concatenate A then B, stable argsort by start only; changes[i] = source differs
    between sorted run i and i+1.  For each consecutive pair (run1=i, run2=i+1):
    check_run is the run1 of the latest change at or before i; pairs before the
    first change are skipped; if check_run.end < run2.start skip, else add
    min(ends) - max(starts).  For well-formed (sorted, disjoint) RLEs this is the
    true intersection; malformed ones (xz tracker wrap bug) get exactly this sweep.
    """
    ra = np.stack([starts_a, starts_a + runs_a], axis=1)
    rb = np.stack([starts_b, starts_b + runs_b], axis=1)
    merged = np.concatenate([ra, rb], axis=0).astype(np.int64)
    ids = np.concatenate([np.zeros(len(ra), np.int8), np.ones(len(rb), np.int8)])
    order = np.argsort(merged[:, 0], kind='stable')
    merged = merged[order]
    ids = ids[order]
    if len(merged) < 2:
        return 0
    changes = ids[:-1] != ids[1:]
    idx = np.arange(len(changes))
    last = np.maximum.accumulate(np.where(changes, idx, -1))
    valid = last >= 0
    chk = merged[np.where(valid, last, 0)]
    run2 = merged[1:]
    ok = valid & ~(chk[:, 1] < run2[:, 0])
    contrib = np.minimum(chk[:, 1], run2[:, 1]) - np.maximum(chk[:, 0], run2[:, 0])
    return int(contrib[ok].sum())


def rle_iou(starts_a, runs_a, starts_b, runs_b, return_intersection=False):
    """array_utils.py:405-429 -- python-float (fp64) division of integers."""
    inter = rle_intersection(starts_a, runs_a, starts_b, runs_b)
    union = int(np.sum(runs_a)) + int(np.sum(runs_b)) - inter
    iou = np.float64(inter) / np.float64(union)
    return (iou, inter) if return_intersection else iou


def rle_ioa(starts_a, runs_a, starts_b, runs_b, return_intersection=False):
    """array_utils.py:431-455 -- area is that of B."""
    inter = rle_intersection(starts_a, runs_a, starts_b, runs_b)
    ioa = np.float64(inter) / np.float64(int(np.sum(runs_b)))
    return (ioa, inter) if return_intersection else ioa


# ----------------------------------------------------------------------------- voting / joining
def concat_sort_ranges(list_of_ranges):
    """array_utils.py:625-632 -- concat non-empty lists, stable sort by start."""
    lst = [np.asarray(r) for r in list_of_ranges if len(r) > 0]
    ranges = np.concatenate(lst, axis=0)
    return ranges[np.argsort(ranges[:, 0], kind='stable')]


def _split_by_votes(start, votes, thr):
    """array_utils.py:457-497 -- maximal sub-ranges of [start, start+len(votes)) with votes >= thr."""
    ok = np.asarray(votes) >= thr
    if not ok.any():
        return []
    d = np.diff(np.concatenate([[0], ok.astype(np.int8), [0]]))
    s = np.where(d == 1)[0] + start
    e = np.where(d == -1)[0] + start
    return [[int(a), int(b)] for a, b in zip(s, e)]


def rle_voting(ranges, vote_thr=2):
    """array_utils.py:539-601 (+ extend_range :499-537), literal sweep.

    Walks consecutive pairs of start-sorted ranges keeping a running range and a
    per-index vote array; a range that *touches* the running range extends it but
    adds votes only where it overlaps.  With fewer than two ranges the loop body
    never runs and the result is empty.
    """
    assert vote_thr > 1, "For vote_thr of 1 use join_ranges instead!"
    ranges = np.asarray(ranges, dtype=np.int64)
    voted = []
    run_s = run_e = None
    votes = None
    for i in range(len(ranges) - 1):
        r1, r2 = ranges[i], ranges[i + 1]
        if run_s is None:
            run_s, run_e = int(r1[0]), int(r1[1])
            votes = np.ones(run_e - run_s, dtype=np.int64)
        if run_e < r2[0]:
            voted.extend(_split_by_votes(run_s, votes, vote_thr))
            run_s = run_e = votes = None
        else:
            first = int(r2[0]) - run_s
            last = len(votes)
            end_off = int(r2[1]) - run_e
            if end_off > 0:
                votes = np.concatenate([votes, np.ones(end_off, dtype=np.int64)])
                run_e = int(r2[1])
            elif end_off < 0:
                last += end_off
            votes[first:last] += 1
    if run_s is not None:
        voted.extend(_split_by_votes(run_s, votes, vote_thr))
    return voted


def _join_ranges(ranges):
    """array_utils.py:634-663 -- union of start-sorted ranges (touching ranges merge).

    A single input range raises UnboundLocalError in the reference (:659-661,
    `range2` never bound); reproduced.
    """
    ranges = np.asarray(ranges, dtype=np.int64)
    if len(ranges) < 2:
        raise UnboundLocalError("local variable 'range2' referenced before assignment")
    joined = []
    run = None
    for i in range(len(ranges) - 1):
        r1, r2 = ranges[i], ranges[i + 1]
        if run is None:
            run = [int(r1[0]), int(r1[1])]
        if run[1] >= r2[0]:
            run[1] = max(run[1], int(r2[1]))
        else:
            joined.append(run)
            run = None
    if run is not None:
        joined.append(run)
    else:
        joined.append([int(ranges[-1][0]), int(ranges[-1][1])])
    return joined


def join_ranges(list_of_ranges):
    """array_utils.py:665-671"""
    lst = [r for r in list_of_ranges if len(r) > 0]
    return np.array(_join_ranges(concat_sort_ranges(lst)))


def vote_by_ranges(list_of_ranges, vote_thr=2):
    """array_utils.py:603-615"""
    lst = [r for r in list_of_ranges if len(r) > 0]
    if vote_thr == 1:
        return join_ranges(lst)
    if len(lst) >= vote_thr:
        return np.array(rle_voting(concat_sort_ranges(lst), vote_thr))
    return np.array([])


def merge_rles(starts_a, runs_a, starts_b=None, runs_b=None):
    """array_utils.py:690-723"""
    lst = [np.stack([starts_a, starts_a + runs_a], axis=1)]
    if starts_b is not None and runs_b is not None:
        lst.append(np.stack([starts_b, starts_b + runs_b], axis=1))
    joined = ranges_to_rle(join_ranges(lst))
    return joined[:, 0], joined[:, 1]


def numpy_fill_instances(volume, instances):
    """array_utils.py:725-737 -- paint runs into the raveled volume, dict order."""
    shape = volume.shape
    flat = volume.reshape(-1)
    for instance_id, attrs in instances.items():
        starts = attrs['starts']
        ends = starts + attrs['runs']
        for s, e in zip(starts, ends):
            flat[s:e] = instance_id
    return flat.reshape(shape)
