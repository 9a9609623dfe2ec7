"""Orchestration of 3D inference, reference names (``empanada/inference/patterns.py`` __all__ :15-31)
plus the MI355X-native whole-stack path.

Per-slice protocol (drop-in, same call sites as scripts/pdl_inference3d.py:140-205):
  create_matchers, create_axis_trackers, apply_matchers, forward_matching, backward_matching,
  update_trackers, finish_tracking, apply_filters, get_axis_trackers_by_class,
  create_instance_consensus, create_semantic_consensus, fill_volume, fill_panoptic_volume,
  all_gather, harden_seg, get_panoptic_seg, forward_multigpu.

Whole-stack protocol (what bench.py and the drivers use): ``track_stack`` takes the panoptic label
stack of one plane as it sits in HBM, extracts runs / components / slice-to-slice overlaps with three
kernel groups and runs the label-propagation chain (forward + backward matching) on the O(#objects)
tables; the result is identical to forward_matching + backward_matching + update_trackers on the same
slices (tests/test_pipeline_gpu.py).
"""
import ctypes
import functools
import multiprocessing

import numpy as np
import torch
import torch.distributed as dist
from scipy.optimize import linear_sum_assignment

from .. import _hip
from ..array_utils import numpy_fill_instances
from ..consensus import merge_objects_from_trackers, merge_semantic_from_trackers
from ..zarr_utils import zarr_fill_instances
from . import filters
from .deferred import LazyFinal, LazySeg
from .engines import _MedianQueue
from .matcher import RLEMatcher
from .postprocess import merge_semantic_and_instance
from .rle import pan_seg_to_rle_seg
from .tracker import InstanceTracker

__all__ = [
    'create_matchers', 'create_axis_trackers', 'apply_matchers', 'forward_matching', 'backward_matching',
    'update_trackers', 'finish_tracking', 'apply_filters', 'get_axis_trackers_by_class',
    'create_instance_consensus', 'create_semantic_consensus', 'fill_volume', 'fill_panoptic_volume',
    'all_gather', 'forward_multigpu', 'harden_seg', 'get_panoptic_seg',
    'track_stack', 'fill_volume_device', 'tables_from_stack', 'chain_from_tables',
]


# ----------------------------------------------------------------------------- reference protocol
def create_matchers(thing_list, label_divisor, merge_iou_thr, merge_ioa_thr):
    """patterns.py:33-39"""
    return [RLEMatcher(thing_class, label_divisor, merge_iou_thr, merge_ioa_thr) for thing_class in thing_list]


def create_axis_trackers(axes, class_labels, label_divisor, shape):
    """patterns.py:41-53"""
    return {axis_name: [InstanceTracker(class_id, label_divisor, shape, axis_name) for class_id in class_labels]
            for axis_name in axes}


def apply_matchers(rle_seg, matchers):
    """patterns.py:55-66.  A handle of a deferred stack (inference/deferred.py) is recorded and returned as it is."""
    if isinstance(rle_seg, LazySeg):
        session = rle_seg._s
        if not session.lazy_apply(rle_seg, matchers):
            _apply_matchers_now(rle_seg._force(), matchers)
            session.note_applied_now(rle_seg)
        return rle_seg
    return _apply_matchers_now(rle_seg, matchers)


def _apply_matchers_now(rle_seg, matchers):
    for matcher in matchers:
        class_id = matcher.class_id
        if matcher.target_rle is None:
            matcher.initialize_target(rle_seg[class_id])
        else:
            rle_seg[class_id] = matcher(rle_seg[class_id])
    return rle_seg


def _end_of_stream(item):
    """the sentinel the feeding loops put last: a string, or a tuple led by one (patterns.py:79-82, 316-318)"""
    return isinstance(item, str) or (isinstance(item, tuple) and len(item) > 0 and isinstance(item[0], str))


def _gpu_process_entry(fn):
    """The reference starts its matcher with ``mp.Process(target=forward_matching, ...)`` from a process that already
    holds the model on the GPU (scripts/pdl_inference3d.py:143-151) -- a FORK on Linux.  The reference's matcher is numpy;
    this one needs the GPU, and a forked child of a process that has initialised HIP cannot use it.  In that case the entry
    point starts a SPAWNED process (own interpreter, own HIP context) that runs the loop, and stays behind as a relay:
    items of the caller's queue go on to the worker's queue (a fork-context queue cannot be handed to a spawned process:
    its semaphores are unnamed), the worker's answer goes back through the caller's pipe.  The script runs unchanged; a
    few seconds of start-up per plane and one more pickle per image.  Anywhere else (main process, thread, spawned
    process) the function runs in place."""
    import inspect
    signature = inspect.signature(fn)

    @functools.wraps(fn)
    def entry(*args, **kwargs):
        if not torch.cuda._is_in_bad_fork():
            return fn(*args, **kwargs)
        ctx = multiprocessing.get_context('spawn')
        bound = signature.bind(*args, **kwargs)              # queue and pipe end by NAME, however they were passed
        queue, matcher_in = bound.arguments['queue'], bound.arguments['matcher_in']
        work_queue = ctx.Queue()
        answer_out, answer_in = ctx.Pipe()
        bound.arguments['queue'], bound.arguments['matcher_in'] = work_queue, answer_in
        proc = ctx.Process(target=entry, args=bound.args, kwargs=bound.kwargs)
        proc.start()
        answer_in.close()
        while True:
            item = queue.get()
            work_queue.put(item)
            if _end_of_stream(item):
                break
        try:
            answer = answer_out.recv()
        except EOFError:
            proc.join()
            raise RuntimeError(f"{fn.__name__}: the spawned matcher process died (exit code {proc.exitcode})")
        matcher_in.send(answer)
        matcher_in.close()
        proc.join()
        return None
    return entry


@_gpu_process_entry
def forward_matching(matchers, queue, rle_stack, matcher_in, labels, label_divisor, thing_list):
    """patterns.py:68-100 -- consumer loop of the matcher process (mp.Queue in, mp.Pipe out)."""
    while True:
        pan_seg = queue.get()
        if pan_seg is None:
            continue
        elif type(pan_seg) == str:
            break
        else:
            rle_seg = pan_seg_to_rle_seg(pan_seg, labels, label_divisor, thing_list, force_connected=True)
            rle_seg = apply_matchers(rle_seg, matchers)
            rle_stack.append(rle_seg)
    matcher_in.send([rle_stack])
    matcher_in.close()


def backward_matching(rle_stack, matchers, axis_len):
    """patterns.py:102-121.  Over the handles of one deferred stack (all of them, in order, straight from
    apply_matchers) it yields handles; the label propagation then runs once, over the whole stack, when the
    trackers are finished."""
    first = rle_stack[0] if len(rle_stack) else None
    if isinstance(first, LazySeg) and first._s.lazy_backward(rle_stack, matchers, axis_len):
        session = first._s
        for rev_idx in range(axis_len - 1, -1, -1):
            yield rev_idx, (LazyFinal(session, rev_idx) if session.bwd == 'lazy' else session.bwd_real[rev_idx])
        return
    # computed on the spot: what apply_matchers recorded for handles in the stack is filed first, so that the
    # matchers are where the reference's forward pass leaves them BEFORE they are reset for the backward pass
    for session in {id(x._s): x._s for x in rle_stack if isinstance(x, LazySeg)}.values():
        session._forward_now()
    yield from _backward_matching_now(rle_stack, matchers, axis_len)


def _backward_matching_now(rle_stack, matchers, axis_len):
    for matcher in matchers:
        matcher.target_rle = None
        matcher.assign_new = False
    for rev_idx in np.arange(0, axis_len)[::-1]:
        rev_idx = rev_idx.item()
        rle_seg = apply_matchers(rle_stack[rev_idx], matchers)
        yield rev_idx, rle_seg


def update_trackers(rle_seg, index, trackers, *unused):
    """patterns.py:123-134 (scripts/pdl_inference3d.py:191 passes two extra arguments; tolerated)."""
    for tracker in trackers:
        tracker.update(rle_seg[tracker.class_id], index)


def finish_tracking(trackers):
    """patterns.py:136-139"""
    for tracker in trackers:
        tracker.finish()


def apply_filters(tracker, filters_dict):
    """patterns.py:141-152"""
    if filters_dict is not None:
        for filt in filters_dict:
            kwargs = {k: v for k, v in filt.items() if k != 'name'}
            filters.__dict__[filt['name']](tracker, **kwargs)


def get_axis_trackers_by_class(trackers, class_id):
    """patterns.py:154-166"""
    return [tr for axis_trackers in trackers.values() for tr in axis_trackers if tr.class_id == class_id]


def create_instance_consensus(class_trackers, pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False):
    """patterns.py:168-186"""
    t0 = class_trackers[0]
    consensus_tracker = InstanceTracker(t0.class_id, t0.label_divisor, t0.shape3d, 'xy')
    consensus_tracker.instances = merge_objects_from_trackers(class_trackers, pixel_vote_thr, cluster_iou_thr, bypass)
    return consensus_tracker


def create_semantic_consensus(class_trackers, pixel_vote_thr=2):
    """patterns.py:188-202"""
    t0 = class_trackers[0]
    consensus_tracker = InstanceTracker(t0.class_id, t0.label_divisor, t0.shape3d, 'xy')
    consensus_tracker.instances = merge_semantic_from_trackers(class_trackers, pixel_vote_thr)
    return consensus_tracker


def fill_volume(volume, instances, processes=4):
    """patterns.py:204-213 -- numpy volumes are painted whole on the GPU, chunked arrays chunk by chunk."""
    if isinstance(volume, np.ndarray):
        numpy_fill_instances(volume, instances)
    elif hasattr(volume, 'chunks') and hasattr(volume, 'shape'):
        zarr_fill_instances(volume, instances, processes)
    else:
        raise Exception(f'Unknown volume type of {type(volume)}')


def fill_panoptic_volume(volume, trackers, processes=4):
    """patterns.py:215-220"""
    for tracker in trackers:
        fill_volume(volume, tracker.instances, processes)


def all_gather(tensor, group=None):
    """patterns.py:226-240 -- RCCL all_gather of same-shape tensors (backend 'nccl' is RCCL on ROCm)."""
    if not dist.is_available() or not dist.is_initialized():
        return [tensor]
    tensor_list = [torch.zeros_like(tensor) for _ in range(dist.get_world_size())]
    dist.all_gather(tensor_list, tensor, group=group)
    return tensor_list


def harden_seg(sem, confidence_thr):
    """patterns.py:242-251 -> (N,1,H,W) int64 (emp_harden)."""
    _hip.require_gpu()
    sem = sem.float().cuda() if not sem.is_cuda else sem.float()
    N, C, H, W = sem.shape
    out = torch.empty((N, H, W), dtype=torch.uint8, device=sem.device)
    for n in range(N):
        one = sem[n:n + 1].contiguous()
        _hip.call('emp_harden', one.data_ptr(), 1, C, H * W, float(confidence_thr), out[n].data_ptr(), _hip.stream())
    return out[:, None].long()


def get_panoptic_seg(sem, instance_cells, label_divisor, thing_list, stuff_area=32, void_label=0):
    """patterns.py:253-277"""
    sem = sem.cuda() if not sem.is_cuda else sem
    instance_cells = instance_cells.cuda() if not instance_cells.is_cuda else instance_cells
    instance_seg = torch.zeros_like(sem)
    for thing_class in thing_list:
        instance_seg[sem == thing_class] = 1
    instance_seg = (instance_seg * instance_cells).long()
    return merge_semantic_and_instance(sem, instance_seg, label_divisor, thing_list, stuff_area, void_label)


@_gpu_process_entry
def forward_multigpu(matchers, queue, rle_stack, matcher_in, confidence_thr, median_kernel_size, labels,
                     label_divisor, thing_list, stuff_area=32, void_label=0):
    """patterns.py:279-350 -- median queue + panoptic post-processing + RLE + forward matching on rank 0."""
    median_queue = _MedianQueue(median_kernel_size)

    def _consume(sem, cells):
        sem = harden_seg(sem, confidence_thr)
        pan_seg = get_panoptic_seg(sem, cells, label_divisor, thing_list, stuff_area, void_label)
        rle_seg = pan_seg_to_rle_seg(pan_seg.squeeze(), labels, label_divisor, thing_list, force_connected=True)
        rle_stack.append(apply_matchers(rle_seg, matchers))

    while True:
        sem, cells = queue.get()
        if isinstance(sem, str):
            break
        median_queue.enqueue({'sem': sem, 'cells': cells})
        median_out = median_queue.get_next(keys=['sem'])
        if median_out is not None:
            _consume(median_out['sem'], median_out['cells'])
    for qout in median_queue.end():
        _consume(qout['sem'], qout['cells'])
    matcher_in.send([rle_stack])
    matcher_in.close()


# ----------------------------------------------------------------------------- whole-stack path
class _Inst:
    """Ordered instances of one slice for one class, as parallel arrays: labels (int64, dict order),
    comps (flat component ids) with seg (start of each instance's components), areas (int64)."""
    __slots__ = ('labels', 'comps', 'seg', 'areas')

    def __init__(self, labels, comps, seg, areas):
        self.labels, self.comps, self.seg, self.areas = labels, comps, seg, areas

    def __len__(self):
        return len(self.labels)


class _ClassChain:
    """Forward + backward label propagation for one thing class on component tables.

    Every instance of a slice is a union of that slice's connected components, so the intersection of
    two instances is the sum of the component-to-component overlaps the GPU already produced
    (emp_runs_overlap_next); IoU/IoA matrices, the Hungarian step and the labelling rule are the
    reference's (matcher.py:193-224, 292-319).  Box screening (matcher.py:199) only skips pairs whose
    intersection is zero, so it does not change the matrices.
    """

    def __init__(self, class_id, label_divisor, merge_iou_thr, merge_ioa_thr):
        self.class_id = class_id
        self.iou_thr = merge_iou_thr
        self.ioa_thr = merge_ioa_thr
        self.next_label = class_id * label_divisor + 1

    def _match(self, target, match, inter, assign_new):
        """target/match: _Inst; inter: (len(target), len(match)) int64 instance intersections.
        Returns the relabelled (and possibly merged) match instances as a new _Inst."""
        nt, nm = len(target), len(match)
        if nm == 0:
            return match
        new_labels = None
        if nt > 0:
            nzr, nzc = np.nonzero(inter)
            iou = np.zeros(inter.shape, dtype='float')
            ioa = np.zeros(inter.shape, dtype=np.float32)
            iv = inter[nzr, nzc]
            iou[nzr, nzc] = iv / (target.areas[nzr] + match.areas[nzc] - iv)
            ioa[nzr, nzc] = iv / match.areas[nzc]
            if len(nzr) and (np.bincount(nzr, minlength=nt).max() > 1 or np.bincount(nzc, minlength=nm).max() > 1):
                rows, cols = linear_sum_assignment(iou, maximize=True)
                keep = iou[rows, cols] >= self.iou_thr
                rows, cols = rows[keep], cols[keep]
            else:
                # at most one overlap per row and column: the maximum-weight assignment is forced to
                # contain every positive entry (any alternative has a smaller sum), so the pairs that
                # survive the IoU filter are exactly the positive entries above the threshold
                keep = iou[nzr, nzc] >= self.iou_thr
                rows, cols = nzr[keep], nzc[keep]
            new_labels = np.full(nm, -1, dtype=np.int64)
            new_labels[cols] = target.labels[rows]
            un = np.flatnonzero(new_labels < 0)
            if len(un):
                ioa_max = ioa[:, un].max(axis=0)
                merge = ioa_max >= self.ioa_thr
                new_labels[un[merge]] = target.labels[ioa[:, un[merge]].argmax(axis=0)]
                rest = un[~merge]
            else:
                rest = un
        else:
            new_labels = np.full(nm, -1, dtype=np.int64)
            rest = np.arange(nm)
            if 0 >= self.ioa_thr:            # ioa_max = 0 (matcher.py:303) passes a non-positive threshold and the
                raise ValueError("attempt to get argmax of an empty sequence")     # reference fails on the empty argmax
        if len(rest):
            if assign_new:
                new_labels[rest] = self.next_label + np.arange(len(rest))
                self.next_label += len(rest)
            else:
                new_labels[rest] = match.labels[rest]
        # instances that received the same label are merged, in order of first appearance
        uniq, first, inv = np.unique(new_labels, return_index=True, return_inverse=True)
        if len(uniq) == nm:
            return _Inst(new_labels, match.comps, match.seg, match.areas)
        order = np.argsort(first, kind='stable')             # groups in order of first appearance
        rank = np.empty(len(uniq), dtype=np.int64)
        rank[order] = np.arange(len(uniq))
        grp = rank[inv]                                       # group index of every match instance
        sizes = np.diff(np.concatenate([match.seg, [len(match.comps)]]))
        inst_order = np.argsort(grp, kind='stable')           # instances by group, original order inside
        comp_src = np.concatenate([np.arange(match.seg[i], match.seg[i] + sizes[i]) for i in inst_order])
        gsizes = np.bincount(grp, weights=sizes, minlength=len(uniq)).astype(np.int64)
        seg = np.concatenate([[0], np.cumsum(gsizes)[:-1]]).astype(np.int64)
        areas = np.bincount(grp, weights=match.areas, minlength=len(uniq)).astype(np.int64)
        return _Inst(uniq[order], match.comps[comp_src], seg, areas)

    def run(self, slices, pair_matrix):
        """slices[t]: _Inst of the slice's connected components (ascending cc label = dict order);
        pair_matrix(t) -> (n_t, n_t1) int64 overlaps between the components of slice t and t+1, indexed by the
        components' positions in slices[t] / slices[t+1].  pos[comp] gives that position.
        Returns the per-slice instances after the backward pass (patterns.py:102-121 semantics)."""
        n = len(slices)
        if n == 0:
            return []
        pos = self.pos

        def inter_of(M, row_inst, col_inst):
            """sum the component overlaps M over the components of each row / column instance"""
            R = M[pos[row_inst.comps]]
            if len(row_inst.seg) != len(row_inst.comps):
                R = np.add.reduceat(R, row_inst.seg, axis=0)
            C = R[:, pos[col_inst.comps]]
            if len(col_inst.seg) != len(col_inst.comps):
                C = np.add.reduceat(C, col_inst.seg, axis=1)
            return C

        fwd = [None] * n
        fwd[0] = slices[0]
        if len(fwd[0]) > 0:
            self.next_label = int(fwd[0].labels.max()) + 1
        mats = [None] * n
        for t in range(1, n):
            M = pair_matrix(t - 1)
            mats[t - 1] = M
            if len(fwd[t - 1]) and len(slices[t]):
                inter = inter_of(M, fwd[t - 1], slices[t])
            else:
                inter = np.zeros((len(fwd[t - 1]), len(slices[t])), dtype=np.int64)
            fwd[t] = self._match(fwd[t - 1], slices[t], inter, True)
        bwd = [None] * n
        bwd[n - 1] = fwd[n - 1]
        if len(bwd[n - 1]) > 0:
            self.next_label = int(bwd[n - 1].labels.max()) + 1
        for t in range(n - 2, -1, -1):
            if len(bwd[t + 1]) and len(fwd[t]):
                inter = inter_of(mats[t].T, bwd[t + 1], fwd[t])      # rows: slice t+1 groups, cols: slice t groups
            else:
                inter = np.zeros((len(bwd[t + 1]), len(fwd[t])), dtype=np.int64)
            bwd[t] = self._match(bwd[t + 1], fwd[t], inter, False)
        return bwd


def tables_from_stack(pan, labels, thing_list, label_divisor):
    """GPU half of track_stack: run table + connected components (emp_runs_*) and the overlaps between
    consecutive slices (emp_runs_overlap_next) of a (D,H,W) uint32 label stack.
    Returns (RunTable on the device, dict of host tables):
      c_slice, c_label (cc label), c_area, c_box (n,4), c_cls (class of the original value), trip (k,3)."""
    labels = list(labels)
    D = pan.shape[0]
    table = _hip.extract_runs(pan, label_divisor, [l for l in labels if l in list(thing_list)])
    nc = table.n_comp
    trip = _hip.overlap_next(table, label_divisor).cpu().numpy() if D > 1 and nc else np.zeros((0, 3), np.int32)
    r_val = table.r_val.cpu().numpy()
    host = {
        'c_slice': table.c_slice.cpu().numpy().astype(np.int64),
        'c_label': table.c_label.cpu().numpy(),
        'c_area': table.c_area.cpu().numpy(),
        'c_box': table.c_box.cpu().numpy(),
        'c_cls': (r_val[table.c_first.cpu().numpy()].astype(np.int64) // label_divisor) if nc else np.zeros(0, np.int64),
        'trip': trip.astype(np.int64),
    }
    return table, host


_LSAP_FN = ctypes.CFUNCTYPE(ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.c_int64, ctypes.c_int64,
                            ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64))


def _lsap_callback(iou_ptr, n_rows, n_cols, rows_out, cols_out):
    """scipy's Hungarian step for emp_chain_class (the routine the reference calls, matcher.py:213)"""
    try:
        iou = np.ctypeslib.as_array(iou_ptr, shape=(n_rows, n_cols))
        rows, cols = linear_sum_assignment(iou, maximize=True)
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        cols = np.ascontiguousarray(cols, dtype=np.int64)
        ctypes.memmove(rows_out, rows.ctypes.data, rows.nbytes)
        ctypes.memmove(cols_out, cols.ctypes.data, cols.nbytes)
        return len(rows)
    except Exception:                        # never unwind through the C frame
        return -1


_LSAP_C = _LSAP_FN(_lsap_callback)


def _i64p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def chain_from_tables(host, D, labels, thing_list, label_divisor, merge_iou_thr=0.25, merge_ioa_thr=0.25,
                      native=True, lsap='native'):
    """Host half of track_stack: forward + backward label propagation over the component tables of D slices.
    Returns (comp_final (n,) final label per component, first_seen {class: {label: order of first update}}).
    native=True runs the loop in C++ (emp_chain_class); native=False is the numpy statement of the same rules
    (_ClassChain), kept as the executable specification the tests compare the native loop with.
    lsap: 'native' = the library's restatement of scipy's linear_sum_assignment (emp_lsap_maximize, no Python inside
    the loop); 'scipy' = a callback into scipy.optimize.linear_sum_assignment itself, the routine the reference calls
    (matcher.py:213) -- the two are compared by the tests."""
    c_slice, c_label, c_area, c_cls, trip = (host[k] for k in ('c_slice', 'c_label', 'c_area', 'c_cls', 'trip'))
    nc = len(c_slice)
    comp_final = np.zeros(nc, dtype=np.int64)
    first_seen = {}
    for l in labels:
        sel = np.flatnonzero(c_cls == l)
        # per slice: components in ascending label order (= dict order of pan_seg_to_rle_seg)
        sel = sel[np.lexsort((c_label[sel], c_slice[sel]))]
        bounds = np.searchsorted(c_slice[sel], np.arange(D + 1))
        pos = np.zeros(max(nc, 1), dtype=np.int64)
        pos[sel] = np.arange(len(sel)) - bounds[c_slice[sel]]
        is_thing = l in thing_list
        if is_thing:
            # overlap triplets of this class, grouped by the slice of the first component
            ta, tb, tv = trip[:, 0], trip[:, 1], trip[:, 2]
            m = c_cls[ta] == l if len(ta) else np.zeros(0, dtype=bool)
            ta, tb, tv = ta[m], tb[m], tv[m]
            o = np.argsort(c_slice[ta], kind='stable') if len(ta) else np.zeros(0, np.int64)
            ta, tb, tv = ta[o], tb[o], tv[o]
            tb_bounds = np.searchsorted(c_slice[ta], np.arange(D + 1)) if len(ta) else np.zeros(D + 1, np.int64)
            pa, pb = pos[ta], pos[tb]
        else:
            tb_bounds = np.zeros(D + 1, np.int64)
            pa = pb = tv = np.zeros(0, np.int64)
        if native:
            arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in
                    (bounds, c_label[sel], c_area[sel], tb_bounds, pa, pb, tv)]
            out = np.zeros(max(len(sel), 1), dtype=np.int64)
            seen = np.zeros(max(len(sel), 1), dtype=np.int64)
            n_seen = ctypes.c_int64(0)
            rc = _hip.load().emp_chain_class(D, _i64p(arrs[0]), _i64p(arrs[1]), _i64p(arrs[2]), int(is_thing),
                                             _i64p(arrs[3]), _i64p(arrs[4]), _i64p(arrs[5]), _i64p(arrs[6]), int(l),
                                             int(label_divisor), float(merge_iou_thr), float(merge_ioa_thr),
                                             ctypes.cast(_LSAP_C, ctypes.c_void_p) if lsap == 'scipy' else None,
                                             _i64p(out), _i64p(seen),
                                             ctypes.byref(n_seen))
            if rc == 1:
                raise ValueError("attempt to get argmax of an empty sequence")
            if rc != 0:
                raise RuntimeError(f"emp_chain_class failed ({rc})")
            comp_final[sel] = out[:len(sel)]
            first_seen[l] = {int(lab): i for i, lab in enumerate(seen[:n_seen.value].tolist())}
            continue
        slices = []
        for t in range(D):
            cs = sel[bounds[t]:bounds[t + 1]]
            slices.append(_Inst(c_label[cs], cs, np.arange(len(cs), dtype=np.int64), c_area[cs]))
        if is_thing:
            def pair_matrix(t, bounds=bounds, tb_bounds=tb_bounds, pa=pa, pb=pb, tv=tv):
                n0, n1 = bounds[t + 1] - bounds[t], bounds[t + 2] - bounds[t + 1]
                M = np.zeros((n0, n1), dtype=np.int64)
                lo, hi = tb_bounds[t], tb_bounds[t + 1]
                if hi > lo:
                    np.add.at(M, (pa[lo:hi], pb[lo:hi]), tv[lo:hi])
                return M

            chain = _ClassChain(l, label_divisor, merge_iou_thr, merge_ioa_thr)
            chain.pos = pos
            result = chain.run(slices, pair_matrix)
        else:
            result = slices
        lab_seq, seen = [], set()
        for t in range(D - 1, -1, -1):
            inst = result[t]
            if len(inst):
                sizes = np.diff(np.concatenate([inst.seg, [len(inst.comps)]]))
                comp_final[inst.comps] = np.repeat(inst.labels, sizes)
                for lab in inst.labels.tolist():
                    if lab not in seen:
                        seen.add(lab)
                        lab_seq.append(lab)
        first_seen[l] = {lab: i for i, lab in enumerate(lab_seq)}
    return comp_final, first_seen


def track_stack(pan, axis_name, shape3d, labels, thing_list, label_divisor, merge_iou_thr=0.25, merge_ioa_thr=0.25,
                return_table=False, timers=None, as_tracks=False):
    """Panoptic label stack of one plane (D,H,W uint32, device) -> finished InstanceTrackers, one per label.

    Equivalent to, slice by slice: pan_seg_to_rle_seg(force_connected=True) -> apply_matchers (forward) ->
    backward_matching -> update_trackers -> finish_tracking (scripts/pdl_inference3d.py:163-198).
    The runs never leave the device until the trackers are asked for (device_tracks.PlaneTracks.trackers());
    as_tracks=True returns the PlaneTracks itself (what the consensus of the whole-stack path consumes).
    """
    import time
    from . import device_tracks as DT
    _t = [time.perf_counter()]

    def _lap(name):
        if timers is not None:
            now = time.perf_counter()
            timers[name] = timers.get(name, 0.0) + now - _t[0]
            _t[0] = now

    labels = list(labels)
    thing_list = list(thing_list)
    D = pan.shape[0]
    table, host = tables_from_stack(pan, labels, thing_list, label_divisor)
    _lap('runs_cc_overlaps_to_host')
    comp_final, first_seen = chain_from_tables(host, D, labels, thing_list, label_divisor, merge_iou_thr, merge_ioa_thr)
    _lap('matching_chain')
    tracks = DT.plane_tracks(table, host, comp_final, first_seen, axis_name, shape3d, labels, label_divisor)
    _lap('lift_runs')
    if as_tracks:
        return (tracks, table, comp_final) if return_table else tracks
    trackers = tracks.trackers()
    _lap('materialise_trackers')
    return (trackers, table, comp_final) if return_table else trackers


def fill_volume_device(shape3d, trackers, dtype=torch.uint32):
    """Paint finished xy-indexed instances of one or more trackers into a fresh device volume (flat zyx
    indices, emp_fill_runs_u32 / _u8).  Later trackers / instances overwrite earlier ones like
    fill_panoptic_volume (patterns.py:215-220)."""
    _hip.require_gpu()
    n = int(np.prod(shape3d))
    ids, starts, runs, order = [], [], [], []
    for tr in trackers:
        for iid, a in tr.instances.items():
            order.append(np.full(len(a['starts']), len(ids), dtype=np.int32))
            ids.append(int(iid))
            starts.append(np.asarray(a['starts'], dtype=np.int64))
            runs.append(np.asarray(a['runs'], dtype=np.int64))
    if dtype == torch.uint8:
        vol = torch.zeros((n,), dtype=torch.uint8, device='cuda')
        for i, s, r in zip(ids, starts, runs):
            _hip.fill_runs_u8(vol, torch.from_numpy(s).cuda(), torch.from_numpy(r).cuda(), i)
        return vol.reshape(shape3d)
    vol = torch.zeros((n,), dtype=torch.int32, device='cuda').view(torch.uint32)
    if ids and sum(len(s) for s in starts):
        _hip.fill_runs_u32(vol, torch.from_numpy(np.concatenate(starts)).cuda(),
                           torch.from_numpy(np.concatenate(runs)).cuda(),
                           torch.from_numpy(np.concatenate(order)).cuda(),
                           _hip.np_to_dev_u32(np.asarray(ids, dtype=np.int64)))
    return vol.reshape(shape3d)
