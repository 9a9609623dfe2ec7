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
import numpy as np
import torch
import torch.distributed as dist
from scipy.optimize import linear_sum_assignment
from scipy.sparse import coo_matrix

from .. import _hip
from ..array_utils import merge_boxes, numpy_fill_instances, put
from ..consensus import merge_objects_from_trackers, merge_semantic_from_trackers
from ..zarr_utils import zarr_fill_instances
from . import filters
from .engines import _MedianQueue
from .matcher import RLEMatcher, assign_labels
from .postprocess import merge_semantic_and_instance
from .rle import pan_seg_to_rle_seg, rle_seg_to_pan_seg, runs_to_instances
from .tracker import InstanceTracker, to_box3d

__all__ = [
    'create_matchers', 'create_axis_trackers', 'apply_matchers', 'forward_matching', 'backward_matching',
    'update_trackers', 'finish_tracking', 'apply_filters', 'get_axis_trackers_by_class',
    'create_instance_consensus', 'create_semantic_consensus', 'fill_volume', 'fill_panoptic_volume',
    'all_gather', 'forward_multigpu', 'harden_seg', 'get_panoptic_seg',
    'track_stack', 'fill_volume_device',
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
    """patterns.py:55-66"""
    for matcher in matchers:
        class_id = matcher.class_id
        if matcher.target_rle is None:
            matcher.initialize_target(rle_seg[class_id])
        else:
            rle_seg[class_id] = matcher(rle_seg[class_id])
    return rle_seg


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
    """patterns.py:102-121"""
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

    def _match(self, target, match, overlap, assign_new):
        """target/match: ordered {label: (comps, area)}; overlap(ta, mb) -> dense (len(ta), len(mb)) int64 of
        component overlaps (target comps x match comps).  Returns ordered {new_label: (comps, area)}."""
        t_labels = np.array(list(target.keys()))
        m_labels = list(match.keys())
        if len(t_labels) == 0 or len(m_labels) == 0:
            matched = (np.array([]), np.array([]))
            ioa = np.array([])
        else:
            t_comps = [c for comps, _ in target.values() for c in comps]
            m_comps = [c for comps, _ in match.values() for c in comps]
            ov = overlap(t_comps, m_comps)
            t_seg = np.cumsum([0] + [len(c) for c, _ in target.values()])[:-1]
            m_seg = np.cumsum([0] + [len(c) for c, _ in match.values()])[:-1]
            inter = np.add.reduceat(np.add.reduceat(ov, t_seg, axis=0), m_seg, axis=1)
            t_area = np.array([a for _, a in target.values()], dtype=np.int64)
            m_area = np.array([a for _, a in match.values()], dtype=np.int64)
            iou = np.zeros(inter.shape, dtype='float')
            ioa = np.zeros(inter.shape, dtype=np.float32)
            r, c = np.nonzero(inter)
            iou[r, c] = inter[r, c] / (t_area[r] + m_area[c] - inter[r, c])
            ioa[r, c] = inter[r, c] / m_area[c]
            rows, cols = linear_sum_assignment(iou, maximize=True)
            keep = iou[rows, cols] >= self.iou_thr
            rows, cols = rows[keep], cols[keep]
            matched = (t_labels[rows], np.array(m_labels)[cols])
        new_labels, self.next_label = assign_labels(m_labels, t_labels, matched, ioa, self.ioa_thr, assign_new,
                                                    self.next_label)
        out = {}
        for nl, (comps, area) in zip(new_labels, match.values()):
            nl = int(nl)
            if nl not in out:
                out[nl] = (list(comps), area)
            else:
                out[nl] = (out[nl][0] + list(comps), out[nl][1] + area)
        return out

    def run(self, slices, overlap_fwd):
        """slices[t]: ordered {cc_label: ([comp], area)} for this class; overlap_fwd(t, comps_t, comps_t1) gives
        the overlap matrix between comps of slice t (rows) and slice t+1 (cols).  Returns the per-slice
        instance dicts after the backward pass (patterns.py:102-121 semantics)."""
        n = len(slices)
        if n == 0:
            return []
        fwd = [None] * n
        fwd[0] = slices[0]
        if len(fwd[0]) > 0:
            self.next_label = max(fwd[0].keys()) + 1
        for t in range(1, n):
            fwd[t] = self._match(fwd[t - 1], slices[t], lambda ta, mb, t=t: overlap_fwd(t - 1, ta, mb), True)
        bwd = [None] * n
        bwd[n - 1] = fwd[n - 1]
        if len(bwd[n - 1]) > 0:
            self.next_label = max(bwd[n - 1].keys()) + 1
        for t in range(n - 2, -1, -1):
            bwd[t] = self._match(bwd[t + 1], fwd[t], lambda ta, mb, t=t: overlap_fwd(t, mb, ta).T, False)
        return bwd


def track_stack(pan, axis_name, shape3d, labels, thing_list, label_divisor, merge_iou_thr=0.25, merge_ioa_thr=0.25,
                return_table=False, timers=None):
    """Panoptic label stack of one plane (D,H,W uint32, device) -> finished InstanceTrackers, one per label.

    Equivalent to, slice by slice: pan_seg_to_rle_seg(force_connected=True) -> apply_matchers (forward) ->
    backward_matching -> update_trackers -> finish_tracking (scripts/pdl_inference3d.py:163-198).
    """
    import time
    _t = [time.perf_counter()]

    def _lap(name):
        if timers is not None:
            now = time.perf_counter()
            timers[name] = timers.get(name, 0.0) + now - _t[0]
            _t[0] = now

    labels = list(labels)
    thing_list = list(thing_list)
    D, H, W = pan.shape
    table = _hip.extract_runs(pan, label_divisor, [l for l in labels if l in thing_list])
    _lap('extract_runs')
    trip = _hip.overlap_next(table, label_divisor).cpu().numpy() if D > 1 and table.n_comp else np.zeros((0, 3), np.int32)
    nc = table.n_comp
    c_slice = table.c_slice.cpu().numpy()
    c_label = table.c_label.cpu().numpy()
    c_area = table.c_area.cpu().numpy()
    c_box = table.c_box.cpu().numpy()
    r_val = table.r_val.cpu().numpy()
    c_cls = (r_val[table.c_first.cpu().numpy()].astype(np.int64) // label_divisor) if nc else np.zeros(0, np.int64)
    S = coo_matrix((trip[:, 2].astype(np.int64), (trip[:, 0], trip[:, 1])), shape=(max(nc, 1), max(nc, 1))).tocsr()
    _lap('overlaps_and_tables_to_host')

    def overlap_fwd(t, comps_t, comps_t1):
        if not len(comps_t) or not len(comps_t1):
            return np.zeros((len(comps_t), len(comps_t1)), dtype=np.int64)
        return np.asarray(S[comps_t][:, comps_t1].todense(), dtype=np.int64)

    # per slice / class: components in ascending label order (= dict order of pan_seg_to_rle_seg)
    order = np.lexsort((c_label, c_cls, c_slice)) if nc else np.zeros(0, np.int64)
    comp_final = np.zeros(nc, dtype=np.int64)        # final label per component
    first_seen = {l: {} for l in labels}             # label -> order of first tracker.update (slice desc, dict order)
    per_class = {l: [dict() for _ in range(D)] for l in labels}
    for c in order:
        k = int(c_cls[c])
        if k in per_class:
            per_class[k][int(c_slice[c])][int(c_label[c])] = ([int(c)], int(c_area[c]))
    for l in labels:
        if l in thing_list:
            chain = _ClassChain(l, label_divisor, merge_iou_thr, merge_ioa_thr)
            result = chain.run(per_class[l], overlap_fwd)
        else:
            result = per_class[l]
        seq = 0
        for t in range(D - 1, -1, -1):
            for lab, (comps, _) in result[t].items():
                comp_final[comps] = lab
                if lab not in first_seen[l]:
                    first_seen[l][lab] = seq
                    seq += 1

    _lap('matching_chain')
    trackers = _assemble_trackers(table, comp_final, c_slice, c_cls, c_box, first_seen, axis_name, shape3d, labels,
                                  label_divisor)
    _lap('assemble_trackers')
    return (trackers, table, comp_final) if return_table else trackers


def _assemble_trackers(table, comp_final, c_slice, c_cls, c_box, first_seen, axis_name, shape3d, labels,
                       label_divisor):
    """Build the InstanceTracker.instances dicts (tracker.py:61-123 semantics) from the run table and the final
    component labels with vectorised numpy over the O(#runs) table."""
    D, H, W = table.D, table.H, table.W
    r_start = table.r_start.cpu().numpy().astype(np.int64)
    r_len = table.r_len.cpu().numpy().astype(np.int64)
    r_comp = table.r_comp.cpu().numpy()
    trackers = []
    n_runs = len(r_start)
    r_slice = c_slice[r_comp] if n_runs else np.zeros(0, np.int64)
    r_label = comp_final[r_comp] if n_runs else np.zeros(0, np.int64)
    r_cls = c_cls[r_comp] if n_runs else np.zeros(0, np.int64)
    for l in labels:
        tr = InstanceTracker(l, label_divisor, shape3d, axis_name)
        sel = np.flatnonzero(r_cls == l)
        if len(sel):
            sl, lb, st, ln = r_slice[sel], r_label[sel], r_start[sel], r_len[sel]
            # runs of one (label, slice) in start order; slices descending (backward pass order)
            o = np.lexsort((st, -sl, lb))
            sl, lb, st, ln = sl[o], lb[o], st[o], ln[o]
            # merge runs that touch inside one (label, slice): rle_encode / join_ranges result
            brk = np.ones(len(st), dtype=bool)
            brk[1:] = (lb[1:] != lb[:-1]) | (sl[1:] != sl[:-1]) | (st[1:] != st[:-1] + ln[:-1])
            seg = np.flatnonzero(brk)
            st, sl, lb = st[seg], sl[seg], lb[seg]
            ln = np.add.reduceat(ln, seg)
            Z, Y, X = shape3d
            if axis_name == 'xy':
                st3 = st + sl * (H * W)
                ln3 = ln
            elif axis_name == 'xz':           # 2D plane (Z, X): only the run START is mapped (tracker.py:78-82)
                st3 = (st // W) * (Y * X) + sl * X + (st % W)
                ln3 = ln
            else:                              # 2D plane (Z, Y): pixels become unit runs, sorted + re-encoded
                rep = np.repeat(np.arange(len(st)), ln)
                pix = st[rep] + (np.arange(len(rep)) - np.repeat(np.cumsum(ln) - ln, ln))
                vox = (pix // W) * (Y * X) + (pix % W) * X + sl[rep]
                vlab = lb[rep]
                o2 = np.lexsort((vox, vlab))
                vox, vlab = vox[o2], vlab[o2]
                b2 = np.ones(len(vox), dtype=bool)
                b2[1:] = (vlab[1:] != vlab[:-1]) | (vox[1:] != vox[:-1] + 1)
                s2 = np.flatnonzero(b2)
                st3, lb = vox[s2], vlab[s2]
                ln3 = np.diff(np.concatenate([s2, [len(vox)]]))
            cuts = np.flatnonzero(np.diff(lb)) + 1
            lab_vals = lb[np.concatenate([[0], cuts])]
            st_parts = np.split(st3, cuts)
            ln_parts = np.split(ln3, cuts)
            # boxes: merge of the per-slice 3D boxes of the member components
            csel = np.flatnonzero(c_cls == l)
            boxes = {}
            for c in csel:
                b3 = to_box3d(int(c_slice[c]), tuple(int(v) for v in c_box[c]), axis_name)
                lab = int(comp_final[c])
                boxes[lab] = b3 if lab not in boxes else merge_boxes(b3, boxes[lab])
            inst = {int(lab): {'box': boxes[int(lab)], 'starts': s, 'runs': r}
                    for lab, s, r in zip(lab_vals, st_parts, ln_parts)}
            for lab in sorted(inst, key=lambda k: first_seen[l][k]):
                tr.instances[lab] = inst[lab]
        tr.finished = True
        trackers.append(tr)
    return trackers


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
