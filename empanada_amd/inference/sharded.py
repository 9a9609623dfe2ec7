"""Slice-sharded inference across the GPUs of one node: one process per GPU, torch.distributed over RCCL (backend
'nccl' on ROCm), contiguous slice blocks per rank -- of every plane, over ONE shared volume.

What the reference does (scripts/inference3d_multigpu.py:353-375, inference/patterns.py:226-240): strided slices per
rank, an all_gather of `sem` and `cells` after EVERY slice, and one CPU process on rank 0 that does median -> fusion ->
CC -> RLE -> matching for the whole stack.

What this module does instead (MI355X-first; no rank is special, nothing is pickled, no pixels through the host):
  1. every rank runs the model on its own contiguous block of slices (no communication);
  2. the recursive median (serial in z by definition, engines.py:68-90) is handed over from rank to rank: rank r
     uses the last ks//2 FILTERED probability slices of rank r-1 and the first ks//2 RAW slices of rank r+1 (one
     all-gather of the raw heads + world-1 small broadcasts of filtered tails: collectives on one communicator only),
     filters its block and passes its own tail on.  The chain is serial but each link costs ~0.2 ms; memory stays
     O(block);
  3. centres, grouping, fusion, runs, connected components: local, on the rank's own slices;
  4. a one-slice halo (first label slice of the next rank) gives the overlaps across block borders;
  5. the O(#components) tables are all-gathered as ONE padded int64 tensor and EVERY rank runs the (deterministic)
     label-propagation chain over the whole axis -- replicated host work instead of gather-to-0 + broadcast;
  6. stack mode: each rank paints its own z-slab.  Orthoplane mode: each rank lifts the 3D runs of its slices on the
     device (device_tracks.py); for the consensus the runs of all planes are all-gathered (~20 MB per plane at 1024^3),
     clipped to the rank's z-slab of the OUTPUT volume, and every rank votes, paints and copies out its own slab;
     pair intersections and voted areas are summed with two small all-reduces, the graph logic is replicated.
Host logic is covered on CPU with the gloo backend (tests/test_sharded_gloo.py).
"""
import numpy as np
import torch
import torch.distributed as dist

from .. import _hip
from . import device_tracks as DT
from .patterns import chain_from_tables, tables_from_stack
from .postprocess import centers_batched

__all__ = ['shard_bounds', 'merge_rank_tables', 'filter_labels', 'gather_tables_and_chain', 'chain_over_ranks',
           'median_handover', 'sharded_panoptic_stack', 'sharded_tables', 'fill_slab', 'sharded_stack_volume',
           'track_plane', 'finish_plane', 'gather_plane_runs', 'gather_plane_tracks', 'consensus_volume', 'plane_volume']


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_bounds(n_slices, world):
    """Contiguous blocks whose sizes differ by at most one: rank r owns [b[r], b[r+1])."""
    base, rem = divmod(n_slices, world)
    sizes = [base + (1 if r < rem else 0) for r in range(world)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


# ----------------------------------------------------------------------------- host logic (CPU-testable)
def merge_rank_tables(rank_tables, counts):
    """Stitch per-rank component tables into one table over the whole axis.

    rank_tables[r]: dict with c_slice (local, 0..counts[r]; slice == counts[r] is the halo = first slice of
    rank r+1), c_label, c_area, c_box, c_cls, trip (a, b, overlap with local component indices).
    Halo components are duplicates of the next rank's slice-0 components and are identified with them by
    (class, cc label) -- the labelling of a slice is deterministic.
    Returns (global table dict, own_index list: for every rank the global id of each local component, -1 for
    halo components).
    """
    world = len(rank_tables)
    bounds = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    own_index, offset = [], 0
    for r, t in enumerate(rank_tables):
        own = t['c_slice'] < counts[r]
        idx = np.full(len(own), -1, dtype=np.int64)
        idx[own] = offset + np.arange(int(own.sum()))
        offset += int(own.sum())
        own_index.append(idx)
    cols = {k: [] for k in ('c_slice', 'c_label', 'c_area', 'c_box', 'c_cls')}
    trips = []
    for r, t in enumerate(rank_tables):
        own = own_index[r] >= 0
        cols['c_slice'].append(t['c_slice'][own] + bounds[r])
        for k in ('c_label', 'c_area', 'c_box', 'c_cls'):
            cols[k].append(t[k][own])
        gmap = own_index[r].copy()
        halo = np.flatnonzero(~own)
        if len(halo):
            assert r + 1 < world, "the last rank has no halo slice"
            nxt = rank_tables[r + 1]
            first = np.flatnonzero(nxt['c_slice'] == 0)
            key = {(int(c), int(l)): int(own_index[r + 1][i]) for i, c, l in
                   zip(first, nxt['c_cls'][first], nxt['c_label'][first])}
            for i in halo:
                gmap[i] = key[(int(t['c_cls'][i]), int(t['c_label'][i]))]
        tr = t['trip']
        if len(tr):
            keep = own[tr[:, 0]]                          # pairs (halo, beyond) do not exist; keep own -> own/halo
            tr = tr[keep]
            trips.append(np.stack([gmap[tr[:, 0]], gmap[tr[:, 1]], tr[:, 2]], axis=1))
    out = {k: (np.concatenate(v) if v else np.zeros(0, np.int64)) for k, v in cols.items()}
    if out['c_box'].ndim == 1:
        out['c_box'] = out['c_box'].reshape(-1, 4)
    out['trip'] = np.concatenate(trips).astype(np.int64) if trips else np.zeros((0, 3), np.int64)
    return out, own_index


def filter_labels(host, comp_final, min_size=None, min_span=None):
    """remove_small_objects + remove_pancakes (inference/filters.py:9-43) evaluated on the component tables of
    an xy stack: returns comp_final with the labels of removed instances set to 0."""
    if len(comp_final) == 0 or (min_size is None and min_span is None):
        return comp_final
    # instances of different classes never share a label value except through overflow; key on (class, label)
    key = host['c_cls'] * (int(comp_final.max()) + 1) + comp_final
    uniq, inv = np.unique(key, return_inverse=True)
    drop = np.zeros(len(uniq), dtype=bool)
    if min_size is not None:
        drop |= np.bincount(inv, weights=host['c_area'], minlength=len(uniq)) < min_size
    if min_span is not None:
        big = np.iinfo(np.int64).max
        lo = np.full((len(uniq), 3), big, dtype=np.int64)
        hi = np.zeros((len(uniq), 3), dtype=np.int64)
        b = host['c_box'].astype(np.int64)
        np.minimum.at(lo, inv, np.stack([host['c_slice'], b[:, 0], b[:, 1]], axis=1))
        np.maximum.at(hi, inv, np.stack([host['c_slice'] + 1, b[:, 2], b[:, 3]], axis=1))
        drop |= ((hi - lo) < min_span).any(axis=1)
    out = comp_final.copy()
    out[drop[inv]] = 0
    return out


# ----------------------------------------------------------------------------- collectives on tables
def _staged(t, group):
    """gloo cannot move device tensors: rehearsal runs (several ranks on one GPU) stage through the host"""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def _gather_var(t, group=None):
    """all-gather of tensors whose first dimension differs per rank: one count all-gather + ONE padded
    all_gather_into_tensor; returns the per-rank tensors on t's device."""
    rank, world = _world(group)
    if world == 1:
        return [t]
    nccl = dist.get_backend(group) == 'nccl'
    dev = t.device
    w = t.contiguous() if (nccl or not t.is_cuda) else t.contiguous().cpu()
    if nccl and not w.is_cuda:
        w = w.cuda()
    n = torch.tensor([w.shape[0]], dtype=torch.int64, device=w.device)
    counts = torch.zeros((world,), dtype=torch.int64, device=w.device)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = counts.cpu().tolist()
    mx = max(max(counts), 1)
    pad = torch.zeros((mx,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    pad[:w.shape[0]] = w
    out = torch.empty((world * mx,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return [out[r * mx:r * mx + counts[r]].to(dev) for r in range(world)]


def _all_reduce_sum(a, group=None):
    """int64 numpy vector summed over the ranks"""
    rank, world = _world(group)
    if world == 1 or len(a) == 0:
        return a
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))
    if dist.get_backend(group) == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


_COLS = ('c_slice', 'c_label', 'c_area', 'c_cls')


def _pack_tables(host, n_local):
    """component + overlap tables of one rank as one flat int64 vector: [n_local, n, k, 8 n comp columns, 3 k trips]"""
    n = len(host['c_slice'])
    cols = [np.asarray(host[c], dtype=np.int64).reshape(n, 1) for c in _COLS]
    cols.append(np.asarray(host['c_box'], dtype=np.int64).reshape(n, 4))
    trip = np.asarray(host['trip'], dtype=np.int64).reshape(-1, 3)
    return np.concatenate([np.array([n_local, n, len(trip)], dtype=np.int64), np.concatenate(cols, axis=1).ravel(),
                           trip.ravel()])


def _unpack_tables(flat):
    flat = np.asarray(flat, dtype=np.int64)
    n_local, n, k = (int(x) for x in flat[:3])
    comp = flat[3:3 + 8 * n].reshape(n, 8)
    host = {c: comp[:, i].copy() for i, c in enumerate(_COLS)}
    host['c_box'] = comp[:, 4:8].astype(np.int32)
    host['trip'] = flat[3 + 8 * n:3 + 8 * n + 3 * k].reshape(k, 3).copy()
    return host, n_local


def chain_over_ranks(local_host, n_local, labels, thing_list, label_divisor, merge_iou_thr=0.25, merge_ioa_thr=0.25,
                     group=None):
    """Steps 4-5 of the module docstring.  Every rank passes its local tables (own slices + one halo slice) and
    receives the same result: (tables of the whole axis, final label per component, first_seen, index of each of its
    local components in the whole-axis tables with -1 for halo components, first slice of its block)."""
    rank, world = _world(group)
    labels, thing_list = list(labels), list(thing_list)
    if world == 1:
        final, first_seen = chain_from_tables(local_host, n_local, labels, thing_list, label_divisor, merge_iou_thr,
                                              merge_ioa_thr)
        return local_host, final, first_seen, np.arange(len(final), dtype=np.int64), 0
    parts = _gather_var(torch.from_numpy(_pack_tables(local_host, n_local)), group)
    tables, counts = zip(*[_unpack_tables(p.cpu().numpy()) for p in parts])
    counts = np.array(counts, dtype=np.int64)
    merged, own_index = merge_rank_tables(list(tables), counts)
    final, first_seen = chain_from_tables(merged, int(counts.sum()), labels, thing_list, label_divisor, merge_iou_thr,
                                          merge_ioa_thr)
    return merged, final, first_seen, own_index[rank], int(counts[:rank].sum())


def gather_tables_and_chain(local_host, n_local, labels, thing_list, label_divisor, merge_iou_thr=0.25,
                            merge_ioa_thr=0.25, min_size=None, min_span=None, group=None, return_first_seen=False):
    """Stack mode: chain over the whole axis + size / span filters; every rank gets the final label of each of its
    own components (0 = filtered out; halo components get 0)."""
    merged, final, first_seen, own, _ = chain_over_ranks(local_host, n_local, labels, thing_list, label_divisor,
                                                         merge_iou_thr, merge_ioa_thr, group)
    final = filter_labels(merged, final, min_size, min_span)
    mine = np.zeros(len(own), dtype=np.int64)
    mine[own >= 0] = final[own[own >= 0]]
    return (mine, first_seen) if return_first_seen else mine


# ----------------------------------------------------------------------------- device side
def _all_gather_cat(t, group=None):
    """all_gather of equally shaped device tensors, concatenated along dim 0 (RCCL all-gather over xGMI)."""
    rank, world = _world(group)
    if world == 1:
        return t
    if _staged(t, group):
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(parts, t.contiguous().cpu(), group=group)
        return torch.cat(parts, dim=0).to(t.device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def _broadcast(t, src, group):
    """broadcast of a device tensor in place (staged through the host under gloo)"""
    if _staged(t, group):
        h = t.contiguous().cpu()
        dist.broadcast(h, src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src, group=group)
    return t


def median_handover(prob_local, ks, thr, group=None, median=None):
    """Recursive median + harden of a rank's block of probabilities (D_local, C, H, W) with the exact whole-axis
    result: out[t] = median(out[t-m .. t-1], x[t .. t+m]) (engines.py:68-84), first / last m slices of the AXIS raw.
    The kernel passes the first m slices of whatever stack it is given through unchanged and uses them as history, so
    prepending rank r-1's last m filtered slices (and appending rank r+1's first m raw ones) reproduces the recursion
    bit for bit.  Block sizes may differ; every block must hold at least m slices.  Returns sem (D_local, H, W) u8.

    Communication is collectives on the group's one communicator only (no point-to-point pairs, whose communicators
    RCCL would create lazily inside the timed loop, and no ordering between pairs to reason about):
      * raw halo: ONE all-gather of every rank's first m slices (world x m x C x H x W fp32: 96 MiB at 1024^2, N = 8);
        rank r reads entry r+1;
      * filtered history: the recursion is serial in z by definition, so the tails travel down the ranks one after the
        other: step r = broadcast(src = r) of rank r's last m filtered slices; rank r+1 uses it, the others discard it.
        world-1 broadcasts of m slices, each behind the sender's median kernel (~0.2 ms per link at 1024^2 x 128)."""
    median = median or _hip.median_harden_stack
    rank, world = _world(group)
    m = (int(ks) - 1) // 2
    if world == 1 or m == 0:
        return median(prob_local, ks, thr)
    D = prob_local.shape[0]
    assert D >= m, f"a block of {D} slices is shorter than the median's reach ({m})"
    heads = _all_gather_cat(prob_local[:m].contiguous(), group)            # (world * m, C, H, W)
    right = heads[(rank + 1) * m:(rank + 2) * m] if rank + 1 < world else None
    tail = torch.empty_like(prob_local[:m])
    left = None
    for src in range(rank):                                                # tails of the ranks before this one
        _broadcast(tail, src, group)
        if src == rank - 1:
            left = tail.clone()
    parts = [p for p in (left, prob_local, right) if p is not None]
    sem, filt = median(torch.cat(parts, dim=0), ks, thr, want_prob=True)
    lo = m if rank > 0 else 0
    if rank + 1 < world:
        tail.copy_(filt[lo + D - m:lo + D])
        _broadcast(tail, rank, group)
        for src in range(rank + 1, world - 1):                             # the ranks after this one hand over too
            _broadcast(tail, src, group)
    return sem[lo:lo + D].contiguous()


def sharded_panoptic_stack(sem_prob_local, ctr_hmp_local, offsets_local, *, thing_list, label_divisor=1000,
                           stuff_area=64, void_label=0, nms_threshold=0.1, nms_kernel=7, confidence_thr=0.5,
                           median_kernel_size=3, coarse_boundaries=True, n_classes=None, group=None):
    """panoptic_stack for a rank's block of slices (blocks may differ in size).  Returns pan (D_local, Hp, Wp) uint32."""
    D, C, Hp, Wp = sem_prob_local.shape
    ks = int(median_kernel_size)
    rank, world = _world(group)
    if world == 1:
        assert D >= ks or ks == 1, "stack shorter than the median kernel"
    sem = median_handover(sem_prob_local.float().contiguous(), ks, confidence_thr, group)
    step = 4 if coarse_boundaries else 1
    idx, cnt = centers_batched(ctr_hmp_local, nms_threshold, nms_kernel)
    ids = _hip.group_pixels(idx, cnt, offsets_local.float().contiguous(), step,
                            sem=sem if step == 1 else None, thing_list=thing_list)
    if n_classes is None:
        n_classes = max(2 if C == 1 else C, max(thing_list) + 1)
    return _hip.fuse_panoptic(sem, ids, idx.shape[1], n_classes, thing_list, label_divisor, stuff_area, void_label,
                              up=step)


def sharded_tables(pan_local, labels, thing_list, label_divisor, group=None):
    """Local runs / connected components and, through a one-slice halo (first label slice of the next rank), the
    overlaps across the block border.  Returns (RunTable, host tables)."""
    rank, world = _world(group)
    pan_ext = pan_local
    if world > 1:
        firsts = _all_gather_cat(pan_local[:1].contiguous().view(torch.int32), group).view(torch.uint32)
        if rank + 1 < world:
            pan_ext = torch.cat([pan_local.view(torch.int32), firsts[rank + 1:rank + 2].view(torch.int32)],
                                dim=0).view(torch.uint32)
    return tables_from_stack(pan_ext, labels, thing_list, label_divisor)


def fill_slab(table, final, shape_local):
    """Stack mode, step 6 (device): paint the rank's (D_local, H, W) uint32 slab from its run table and the final
    label of every component (0 = dropped)."""
    D, H, W = shape_local
    vol = torch.zeros((D, H, W), dtype=torch.int32, device=table.r_start.device).view(torch.uint32)
    if table.n_comp:
        _hip.fill_table_u32(vol, table, _hip.np_to_dev_u32(final), slice0=0)
    return vol


def sharded_stack_volume(pan_local, labels, thing_list, label_divisor, merge_iou_thr=0.25, merge_ioa_thr=0.25,
                         min_size=None, min_span=None, group=None):
    """Stack mode, steps 3-6: local runs/CC, halo overlaps, replicated chain, local slab fill.
    Returns the rank's (D_local, H, W) uint32 slab of the labelled volume (device)."""
    table, host = sharded_tables(pan_local, labels, thing_list, label_divisor, group)
    final = gather_tables_and_chain(host, pan_local.shape[0], list(labels), list(thing_list), label_divisor,
                                    merge_iou_thr, merge_ioa_thr, min_size, min_span, group)
    return fill_slab(table, final, tuple(pan_local.shape))


# ----------------------------------------------------------------------------- orthoplane: trackers per plane
def finish_plane(table, host, n_local, axis_name, shape3d, labels, thing_list, label_divisor, merge_iou_thr=0.25,
                 merge_ioa_thr=0.25, inst_base=0, group=None):
    """Everything of a plane after its device tables exist: replicated chain over the whole axis, instance table,
    lift of the rank's own runs into the (Z, Y, X) frame on the device.  Split from track_plane so that a driver can
    queue the next plane's forward on the GPU first.  Returns PlaneTracks (instance table of the WHOLE axis, runs of
    this rank's slices)."""
    merged, final, first_seen, own, slice0 = chain_over_ranks(host, n_local, labels, thing_list, label_divisor,
                                                              merge_iou_thr, merge_ioa_thr, group)
    return DT.plane_tracks(table, merged, final, first_seen, axis_name, shape3d, list(labels), label_divisor,
                           slice0=slice0, inst_base=inst_base, local_comp_index=own)


def track_plane(pan_local, axis_name, shape3d, labels, thing_list, label_divisor, merge_iou_thr=0.25,
                merge_ioa_thr=0.25, inst_base=0, group=None):
    """Orthoplane mode, one plane: every rank passes the panoptic labels of its contiguous block of slices."""
    table, host = sharded_tables(pan_local, list(labels), list(thing_list), label_divisor, group)
    return finish_plane(table, host, pan_local.shape[0], axis_name, shape3d, labels, thing_list, label_divisor,
                        merge_iou_thr, merge_ioa_thr, inst_base, group)


def gather_plane_runs(pt, group=None):
    """all ranks' runs of one plane: (key, ln, n) device arrays, concatenated in rank order ("Collective 2" of SURVEY
    8(e): one count all-gather + one padded all-gather of the flat (key, len) table)"""
    rank, world = _world(group)
    if world == 1:
        return pt.key[:pt.n_runs], pt.ln[:pt.n_runs], pt.n_runs
    local = torch.stack([pt.key[:pt.n_runs], pt.ln[:pt.n_runs]], dim=1)
    allr = torch.cat(_gather_var(local, group), dim=0)
    return allr[:, 0].contiguous(), allr[:, 1].contiguous(), int(allr.shape[0])


def gather_plane_tracks(pt, group=None):
    """PlaneTracks holding every run of the plane on every rank (for callers that want the reference's trackers:
    pt.trackers()); yz runs split at block borders are joined again."""
    rank, world = _world(group)
    if world == 1:
        return pt
    key, ln, n = gather_plane_runs(pt, group)
    out = DT.PlaneTracks(pt.axis, pt.shape3d, pt.labels, pt.label_divisor)
    for name in ('inst_label', 'inst_cls', 'inst_area', 'inst_box', 'alive', 'inst_base'):
        setattr(out, name, getattr(pt, name))
    out.key, out.st, out.ln, out.n_runs = DT.sort_runs(key, ln, n, merge_touching=(pt.axis == 'yz'))
    return out


def consensus_volume(planes, shape3d, labels, thing_list, pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False,
                     min_size=None, min_span=None, group=None, on_degenerate='zero'):
    """Orthoplane mode, last step (scripts/pdl_inference3d.py:200-233): per-plane filters, instance / semantic
    consensus per class, filters again, fill -- sharded by z-slab of the OUTPUT volume.

    planes: {'xy': PlaneTracks, 'xz': ..., 'yz': ...} whose instance indices are disjoint (inst_base) and ascending in
    that order.  Every rank votes on and paints the flat voxel interval [z0 * Y * X, z1 * Y * X) of its own z-slab.

    on_degenerate: the reference's merge functions crash on a degenerate class -- a stuff class voted empty (present in
    fewer planes than pixel_vote_thr: IndexError, consensus.py:331-339) or pixel_vote_thr == 1 with a one-run object
    (UnboundLocalError / ValueError out of join_ranges, array_utils.py:659-671).  `empanada_amd.consensus` keeps those
    exceptions (bug parity of the reference API); this driver runs at the very END of a volume's work, so by default
    ('zero') it warns and gives that class an all-zero volume instead of losing every class; 'raise' re-raises.  The
    conditions are evaluated on all-reduced counts, so every rank takes the same branch.
    Returns ({class: ConsensusResult}, {class: (z1 - z0, Y, X) device slab, uint32 things / uint8 stuff}, (z0, z1))."""
    from .. import consensus as CO
    rank, world = _world(group)
    Z, Y, X = (int(s) for s in shape3d)
    zb = shard_bounds(Z, world)
    z0, z1 = int(zb[rank]), int(zb[rank + 1])
    lo, hi = z0 * Y * X, z1 * Y * X
    order = [planes[a] for a in ('xy', 'xz', 'yz') if a in planes]
    for pt in order:
        if min_size is not None:
            pt.remove_small_objects(min_size)
        if min_span is not None:
            pt.remove_pancakes(min_span)
    # ---- the run store of this rank's slab: every plane's runs (all ranks'), clipped to the slab, sorted
    n_obj = sum(pt.n_inst for pt in order)
    bases = np.cumsum([0] + [pt.n_inst for pt in order])
    for pt, b in zip(order, bases):
        assert pt.inst_base == int(b), "planes must be lifted with consecutive instance bases (xy, xz, yz)"
    keys, lens = [], []
    for pt in order:
        key, ln, n = gather_plane_runs(pt, group)
        if world > 1:
            key, ln, n = DT.clip_runs(key, ln, n, lo, hi)
            if n:
                key, _, ln, n = DT.sort_runs(key, ln, n)           # rank blocks are sorted, their concatenation is not
        keys.append(key[:n])
        lens.append(ln[:n])
    key = torch.cat(keys) if len(keys) > 1 else keys[0]
    ln = torch.cat(lens) if len(lens) > 1 else lens[0]
    n_runs = int(key.numel())
    dev = ln.device
    off = torch.empty((n_obj + 1,), dtype=torch.int64, device=dev)
    _hip.call('emp_track_offsets', _hip._ptr(key) if n_runs else None, n_runs, n_obj, _hip._ptr(off), _hip.stream())
    st = torch.bitwise_and(key, (1 << DT.POS_BITS) - 1)
    store = CO.RunStore(st, ln, off)
    reduce = (lambda a: _all_reduce_sum(a, group)) if world > 1 else None
    src = np.concatenate([np.full(pt.n_inst, i, dtype=np.int64) for i, pt in enumerate(order)])
    boxes = np.concatenate([pt.inst_box for pt in order])
    areas = np.concatenate([pt.inst_area for pt in order])
    alive = np.concatenate([pt.alive for pt in order])
    cls = np.concatenate([pt.inst_cls for pt in order])
    cons, vols = {}, {}
    for class_id in labels:
        nodes = np.flatnonzero(alive & (cls == class_id))
        thing = class_id in thing_list
        try:
            if thing:
                res = CO.consensus_objects(src[nodes], boxes[nodes], areas[nodes], store, len(order), pixel_vote_thr,
                                           cluster_iou_thr, bypass, reduce=reduce, store_index=nodes)
                if min_size is not None:
                    res.remove_small_objects(min_size)
                if min_span is not None:
                    res.remove_pancakes(min_span)
            else:
                res = CO.consensus_semantic(src[nodes], boxes[nodes], store, pixel_vote_thr, reduce=reduce,
                                            store_index=nodes)
        except (IndexError, UnboundLocalError, ValueError) as e:
            if on_degenerate == 'raise':
                raise
            import warnings
            warnings.warn(f"consensus of class {class_id} is degenerate ({type(e).__name__}: {e}; the reference crashes "
                          f"here): the class is written as an all-zero volume", RuntimeWarning)
            res = CO.ConsensusResult(np.zeros((0, 6)), np.zeros(0), None, np.zeros(1))
        vol = torch.zeros(((z1 - z0) * Y * X,), dtype=torch.int32 if thing else torch.uint8, device=dev)
        if thing:
            vol = vol.view(torch.uint32)
        res.paint(vol, lo)
        cons[class_id] = res
        vols[class_id] = vol.reshape(z1 - z0, Y, X)
    return cons, vols, (z0, z1)


def plane_volume(pt, labels, thing_list, min_size=None, min_span=None, group=None):
    """Stack mode with per-class outputs (scripts/pdl_inference3d.py:222-233 with a single axis: the class's tracker IS
    the result): size / span filters on the xy plane's trackers, then every class's instances painted with their own
    labels into the rank's z-slab -- uint32 for thing classes; stuff classes as a uint8 mask (value 1; the reference
    writes the label class * divisor into a uint8 array there, which does not fit).
    Returns ({class: (z1 - z0, Y, X) device slab}, (z0, z1), {class: instances kept})."""
    assert pt.axis == 'xy', "the slices of the xy plane are the z-slabs of the volume"
    rank, world = _world(group)
    Z, Y, X = pt.shape3d
    zb = shard_bounds(Z, world)
    z0, z1 = int(zb[rank]), int(zb[rank + 1])
    if min_size is not None:
        pt.remove_small_objects(min_size)
    if min_span is not None:
        pt.remove_pancakes(min_span)
    dev = pt.ln.device
    n = pt.n_runs
    off = pt.offsets()
    vols, counts = {}, {}
    for c in labels:
        keep = pt.alive & (pt.inst_cls == c)
        counts[c] = int(keep.sum())
        thing = c in thing_list
        vol = torch.zeros(((z1 - z0) * Y * X,), dtype=torch.int32 if thing else torch.uint8, device=dev)
        if n and keep.any():
            st = (pt.st[:n] - z0 * Y * X).contiguous()
            if thing:
                order = torch.empty((n,), dtype=torch.int32, device=dev)
                iota = torch.arange(pt.n_inst, dtype=torch.int32, device=dev)
                _hip.call('emp_track_expand', _hip._ptr(off), _hip._ptr(iota), pt.n_inst, n, _hip._ptr(order), _hip.stream())
                _hip.fill_runs_u32(vol.view(torch.uint32), st, pt.ln[:n].contiguous(), order,
                                   _hip.np_to_dev_u32(np.where(keep, pt.inst_label, 0)))
            else:
                flag = torch.empty((n,), dtype=torch.int32, device=dev)
                kd = torch.from_numpy(keep.astype(np.int32)).to(dev)
                _hip.call('emp_track_expand', _hip._ptr(off), _hip._ptr(kd), pt.n_inst, n, _hip._ptr(flag), _hip.stream())
                sel = flag > 0
                _hip.fill_runs_u8(vol, st[sel].contiguous(), pt.ln[:n][sel].contiguous(), 1)
        vols[c] = (vol.view(torch.uint32) if thing else vol).reshape(z1 - z0, Y, X)
    return vols, (z0, z1), counts
