"""Slice-sharded stack inference across the GPUs of one node: one process per GPU, torch.distributed over
RCCL (backend 'nccl' on ROCm), contiguous slice blocks per rank.

What the reference does (scripts/inference3d_multigpu.py:353-375, inference/patterns.py:226-240): strided
slices per rank, an all_gather of `sem` and `cells` after EVERY slice, and one CPU process on rank 0 that
does median -> fusion -> CC -> RLE -> matching for the whole stack.

What this module does instead (MI355X-first):
  1. every rank runs the model on its own contiguous block of slices (no communication);
  2. ONE all-gather of the semantic probabilities per plane, so that each rank can run the recursive median
     (serial in z by definition, engines.py:68-90) over the whole axis -- elementwise and ~0.1 ms per 256 slices;
  3. centres, grouping, fusion, runs, connected components: local, on the rank's own slices;
  4. a one-slice halo (first label slice of the next rank) gives the overlaps across block borders;
  5. only O(#objects) tables travel to rank 0, which runs the label-propagation chain once and broadcasts
     the final label of every component; each rank paints its own z-slab of the output volume.
No per-slice synchronisation, no pixels through the host.  The host logic (steps 4-5) is plain numpy +
torch.distributed object collectives and is covered on CPU with the gloo backend (tests/test_sharded_gloo.py).
"""
import numpy as np
import torch
import torch.distributed as dist

from .. import _hip
from .patterns import _assemble_trackers, chain_from_tables, merge_partial_trackers, tables_from_stack
from .postprocess import centers_batched

__all__ = ['shard_bounds', 'merge_rank_tables', 'filter_labels', 'gather_tables_and_chain', 'sharded_panoptic_stack',
           'sharded_tables', 'fill_slab', 'sharded_stack_volume', 'partial_trackers', 'sharded_track_plane', 'finish_plane',
           'consensus_volume']


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
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


def gather_tables_and_chain(local_host, n_local, labels, thing_list, label_divisor, merge_iou_thr=0.25,
                            merge_ioa_thr=0.25, min_size=None, min_span=None, group=None, return_first_seen=False):
    """Steps 4-5 of the module docstring.  Every rank passes its local tables (own slices + halo); rank 0 merges
    them, runs the chain over the whole axis and the size/span filters, and every rank receives the final label
    of each of its own components (0 = filtered out; halo components get 0)."""
    rank, world = _world()
    if world == 1:
        final, first_seen = chain_from_tables(local_host, n_local, labels, thing_list, label_divisor, merge_iou_thr,
                                              merge_ioa_thr)
        final = filter_labels(local_host, final, min_size, min_span)
        return (final, first_seen) if return_first_seen else final
    gathered = [None] * world
    dist.all_gather_object(gathered, (local_host, int(n_local)), group=group)
    result = [None]
    first_seen = None
    if rank == 0:
        tables = [g[0] for g in gathered]
        counts = np.array([g[1] for g in gathered], dtype=np.int64)
        merged, own_index = merge_rank_tables(tables, counts)
        final, first_seen = chain_from_tables(merged, int(counts.sum()), labels, thing_list, label_divisor,
                                              merge_iou_thr, merge_ioa_thr)
        final = filter_labels(merged, final, min_size, min_span)
        per_rank = []
        for idx in own_index:
            v = np.zeros(len(idx), dtype=np.int64)
            v[idx >= 0] = final[idx[idx >= 0]]
            per_rank.append(v)
        result = [per_rank]
    dist.broadcast_object_list(result, src=0, group=group)
    final = result[0][rank]
    return (final, first_seen) if return_first_seen else final


# ----------------------------------------------------------------------------- device side
def _all_gather_cat(t, group=None):
    """all_gather of equally shaped device tensors, concatenated along dim 0 (RCCL all-gather over xGMI)."""
    rank, world = _world()
    if world == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        # rehearsal mode (several ranks on one GPU, `EMP_BENCH_BACKEND=gloo`): stage through the host
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(parts, t.contiguous().cpu(), group=group)
        return torch.cat(parts, dim=0).to(t.device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def sharded_panoptic_stack(sem_prob_local, ctr_hmp_local, offsets_local, *, thing_list, label_divisor=1000,
                           stuff_area=64, void_label=0, nms_threshold=0.1, nms_kernel=7, confidence_thr=0.5,
                           median_kernel_size=3, coarse_boundaries=True, n_classes=None, group=None):
    """panoptic_stack for a rank's block of slices.  All ranks must hold the same number of slices (pad the
    volume or use shard sizes that divide it).  Returns pan (D_local, Hp, Wp) uint32."""
    rank, world = _world()
    D, C, Hp, Wp = sem_prob_local.shape
    ks = int(median_kernel_size)
    full = _all_gather_cat(sem_prob_local.float().contiguous(), group)
    assert full.shape[0] >= ks or ks == 1, "stack shorter than the median kernel"
    sem_full = _hip.median_harden_stack(full, ks, confidence_thr)
    sem = sem_full[rank * D:(rank + 1) * D].contiguous()
    step = 4 if coarse_boundaries else 1
    idx, cnt = centers_batched(ctr_hmp_local, nms_threshold, nms_kernel)
    ids = _hip.group_pixels(idx, cnt, offsets_local.float().contiguous(), step,
                            sem=sem if step == 1 else None, thing_list=thing_list)
    if n_classes is None:
        n_classes = max(2 if C == 1 else C, max(thing_list) + 1)
    return _hip.fuse_panoptic(sem, ids, idx.shape[1], n_classes, thing_list, label_divisor, stuff_area, void_label,
                              up=step)


def sharded_tables(pan_local, labels, thing_list, label_divisor, group=None):
    """Stack mode, step 3-4a (device): local runs / connected components and, through a one-slice halo (first
    label slice of the next rank), the overlaps across the block border.  Returns (RunTable, host tables)."""
    rank, world = _world()
    pan_ext = pan_local
    if world > 1:
        firsts = _all_gather_cat(pan_local[:1].contiguous().view(torch.int32), group).view(torch.uint32)
        if rank + 1 < world:
            pan_ext = torch.cat([pan_local.view(torch.int32), firsts[rank + 1:rank + 2].view(torch.int32)],
                                dim=0).view(torch.uint32)
    return tables_from_stack(pan_ext, labels, thing_list, label_divisor)


def fill_slab(table, final, shape_local):
    """Stack mode, step 5 (device): paint the rank's (D_local, H, W) uint32 slab from its run table and the final
    label of every component (0 = dropped)."""
    D, H, W = shape_local
    vol = torch.zeros((D, H, W), dtype=torch.int32, device=table.r_start.device).view(torch.uint32)
    if table.n_comp:
        _hip.fill_table_u32(vol, table, _hip.np_to_dev_u32(final), slice0=0)
    return vol


def sharded_stack_volume(pan_local, labels, thing_list, label_divisor, merge_iou_thr=0.25, merge_ioa_thr=0.25,
                         min_size=None, min_span=None, group=None):
    """Stack mode, steps 3-5: local runs/CC, halo overlaps, global chain on rank 0, local slab fill.
    Returns the rank's (D_local, H, W) uint32 slab of the labelled volume (device)."""
    table, host = sharded_tables(pan_local, labels, thing_list, label_divisor, group)
    final = gather_tables_and_chain(host, pan_local.shape[0], list(labels), list(thing_list), label_divisor,
                                    merge_iou_thr, merge_ioa_thr, min_size, min_span, group)
    return fill_slab(table, final, tuple(pan_local.shape))


# ----------------------------------------------------------------------------- orthoplane: trackers per plane
def partial_trackers(table, host, final_local, axis_name, shape3d, slice0, labels, label_divisor):
    """The rank's share of one plane's trackers: 3D run lists of its own slices (global slice = slice0 + local),
    instances in ascending label order.  Halo components carry label 0 and are skipped."""
    return _assemble_trackers(table, np.asarray(final_local, dtype=np.int64), host['c_slice'], host['c_cls'],
                              host['c_box'], None, axis_name, shape3d, list(labels), label_divisor, slice0=slice0)


def finish_plane(table, host, n_local, axis_name, shape3d, slice0, labels, thing_list, label_divisor,
                 merge_iou_thr=0.25, merge_ioa_thr=0.25, group=None):
    """Host half of sharded_track_plane (everything after the device tables exist): chain on rank 0, partial
    trackers, all-gather of the RLE tables, stitch.  Split out so that a driver can queue the next plane's forward
    on the GPU before calling it."""
    rank, world = _world()
    labels, thing_list = list(labels), list(thing_list)
    final, first_seen = gather_tables_and_chain(host, n_local, labels, thing_list, label_divisor, merge_iou_thr,
                                                merge_ioa_thr, group=group, return_first_seen=True)
    part = partial_trackers(table, host, final, axis_name, shape3d, slice0, labels, label_divisor)
    if world == 1:
        return merge_partial_trackers([part], first_seen, axis_name, shape3d, labels, label_divisor)
    gathered = [None] * world
    dist.all_gather_object(gathered, part, group=group)
    if rank != 0:
        return None
    return merge_partial_trackers(gathered, first_seen, axis_name, shape3d, labels, label_divisor)


def sharded_track_plane(pan_local, axis_name, shape3d, slice0, labels, thing_list, label_divisor, merge_iou_thr=0.25,
                        merge_ioa_thr=0.25, group=None):
    """Orthoplane mode, one plane: every rank passes the panoptic labels of its contiguous block of slices
    (global index of the first one = slice0).  Steps: local runs / CC / halo overlaps, chain over the whole axis on
    rank 0, per-rank partial trackers (the O(#runs) assembly is sharded too), all-gather of the partial per-instance
    3D RLE tables -- the collective SURVEY 8(e) calls "Collective 2" -- and the stitch on rank 0.
    Returns the plane's finished trackers on rank 0 (None on the other ranks)."""
    table, host = sharded_tables(pan_local, list(labels), list(thing_list), label_divisor, group)
    return finish_plane(table, host, pan_local.shape[0], axis_name, shape3d, slice0, labels, thing_list,
                        label_divisor, merge_iou_thr, merge_ioa_thr, group)


def consensus_volume(trackers_by_axis, shape3d, labels, thing_list, pixel_vote_thr=2, cluster_iou_thr=0.75,
                     bypass=False, min_size=None, min_span=None):
    """Orthoplane mode, last step, on the rank that holds the stitched trackers (rank 0): per-plane filters,
    instance / semantic consensus per class, filters again, fill (scripts/pdl_inference3d.py:200-233).
    trackers_by_axis: {'xy': [tracker per label], 'xz': [...], 'yz': [...]}.
    Returns ({class: consensus tracker}, {class: labelled (Z,Y,X) device volume, uint32 for things / uint8 stuff})."""
    from . import filters
    from .patterns import (create_instance_consensus, create_semantic_consensus, fill_volume_device,
                           get_axis_trackers_by_class)
    for trs in trackers_by_axis.values():
        for tr in trs:
            if min_size is not None:
                filters.remove_small_objects(tr, min_size)
            if min_span is not None:
                filters.remove_pancakes(tr, min_span)
    cons, vols = {}, {}
    for class_id in labels:
        cts = get_axis_trackers_by_class(trackers_by_axis, class_id)
        if class_id in thing_list:
            con = create_instance_consensus(cts, pixel_vote_thr, cluster_iou_thr, bypass)
            if min_size is not None:
                filters.remove_small_objects(con, min_size)
            if min_span is not None:
                filters.remove_pancakes(con, min_span)
            vols[class_id] = fill_volume_device(shape3d, [con])
        else:
            con = create_semantic_consensus(cts, pixel_vote_thr)
            vols[class_id] = fill_volume_device(shape3d, [con], dtype=torch.uint8)
        cons[class_id] = con
    return cons, vols
