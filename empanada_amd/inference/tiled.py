"""Tiled inference of planes that are cut into overlapping tiles (BASELINE configs[4]: 2048 x 2048 planes, C = 5).

The reference ships the pieces -- ``Tiler`` (empanada/inference/tile.py:54-194), ``merge_objects_from_tiles`` /
``merge_semantic_from_tiles`` (empanada/consensus.py:471-625) -- and its test shows the call sequence
(tests/test_tiling.py:26-47): every tile's panoptic image -> ``pan_seg_to_rle_seg(..., force_connected=False)`` ->
``translate_rle_seg`` -> per class ``merge_objects_from_tiles`` -> one rle_seg for the plane.  This module is that
sequence as a driver over a whole stack of slices:

  * tile stacks go through the whole-stack post-processing one tile position at a time (`panoptic_stack`: the recursive
    median runs along z inside each tile position, all slices of a tile in five kernel groups);
  * run extraction for ALL slices of a tile is one pass of the run kernels (`_hip.extract_runs`);
  * the tiles' objects of ALL slices and classes are stitched in one go on device tables (`stitch_stack`: lift into the
    image frame, box screen, pair intersections, overlap-region test, ordered fill) -- no Python loop per slice or class;
    `stitch_slice` keeps the reference's per-slice call sequence for the per-slice protocol.
The stitched (D, H, W) stack then enters ``track_stack`` / ``sharded`` like any other plane.
"""
import numpy as np
import torch

from .. import _hip
from ..consensus import merge_objects_from_tiles, merge_semantic_from_tiles
from .postprocess import panoptic_stack
from .rle import stack_to_rle_segs

__all__ = ['stitch_slice', 'stitch_stack', 'tiled_panoptic_stack']

TIMERS = {}           # host seconds spent per stage of tiled_panoptic_stack (accumulated; tools/bench_tiled.py)


def stitch_slice(tile_rle_segs, tiler, labels, thing_list, use_overlap=True):
    """rle_segs of one slice's tiles (tile frame) -> rle_seg of the plane (plane frame): translate, then per class
    merge_objects_from_tiles (things; objects seen in one tile only that lie by more than 10 % inside the overlap
    region are dropped when use_overlap) or merge_semantic_from_tiles (stuff)."""
    moved = [tiler.translate_rle_seg(rs, i) for i, rs in enumerate(tile_rle_segs)]
    out = {}
    for l in labels:
        per_tile = [rs[l] for rs in moved]
        if l in thing_list:
            out[l] = merge_objects_from_tiles(per_tile, tiler.overlap_rle if use_overlap else None)
        else:
            out[l] = merge_semantic_from_tiles(per_tile)
    return out


def _paint(rle_seg, shape, out):
    """rle_seg of one plane -> out (H, W) uint32 device view (zeroed here)"""
    ids, starts, runs, order = [], [], [], []
    for insts in rle_seg.values():
        for object_id, a in insts.items():
            order.append(np.full(len(a['starts']), len(ids), dtype=np.int32))
            ids.append(int(object_id))
            starts.append(np.asarray(a['starts'], dtype=np.int64))
            runs.append(np.asarray(a['runs'], dtype=np.int64))
    flat = out.view(torch.int32).reshape(-1)
    flat.zero_()
    if ids and sum(len(s) for s in starts):
        cat = lambda x, dt: torch.from_numpy(np.concatenate(x).astype(dt)).to(out.device)
        _hip.fill_runs_u32(flat.view(torch.uint32), cat(starts, np.int64), cat(runs, np.int64), cat(order, np.int32),
                           _hip.np_to_dev_u32(np.asarray(ids, dtype=np.int64)))


def _overlap_prefix(tiler, device):
    """prefix[p] = number of overlap-region pixels (Tiler.overlap_rle, tile.py:8-52) with flat image index < p; int32
    (H * W + 1,) on the device, cached on the tiler.  The overlap of a flat range [a, b) with the region is then
    prefix[b] - prefix[a]."""
    cached = getattr(tiler, '_overlap_prefix', None)
    if cached is not None and cached.device == device:
        return cached
    H, W = tiler.image_shape
    assert H * W < 2 ** 31
    mask = torch.zeros((H * W,), dtype=torch.uint8, device=device)
    starts, runs = tiler.overlap_rle
    if len(starts):
        _hip.fill_runs_u8(mask, torch.from_numpy(np.asarray(starts, dtype=np.int64)).to(device),
                          torch.from_numpy(np.asarray(runs, dtype=np.int64)).to(device), 1)
    prefix = torch.zeros((H * W + 1,), dtype=torch.int32, device=device)
    torch.cumsum(mask, 0, dtype=torch.int32, out=prefix[1:])
    tiler._overlap_prefix = prefix
    return prefix


def stitch_stack(tables, tiler, n_slices, labels, thing_list, label_divisor, use_overlap=True, on_single_run='raise',
                 return_rle=False):
    """The tile merge of ALL slices and classes at once, on tables (C5: consensus.py:471-625 + tile.py:122-168).

    tables[i]: `_hip.extract_runs(pan_tile_i, label_divisor, [])` of tile i's (D, th, tw) panoptic stack -- objects are
    the (slice, value) groups, as pan_seg_to_rle_seg(force_connected=False) makes them (tests/test_tiling.py:36-38).
    What the reference does per slice and per class in Python -- translate every object's RLE into the image frame,
    screen box pairs of different tiles, intersect their RLEs, join the connected components of the resulting graph,
    drop single-tile objects lying by more than 10 % in the overlap region, number the survivors from the smallest
    label on -- runs here as: one lift per tile (emp_tile_lift), one sort (emp_track_sort), one box screen
    (emp_box_pairs over (segment, y, x) boxes with segment = (slice, class), so that only objects of one slice and class
    pair up), one intersection launch (emp_rle_pair_intersections), a prefix-table lookup for the overlap region, and
    one ordered fill (emp_fill_runs_u32).  The host sees O(#objects) tables: the node table, the candidate pairs, the
    connected components (scipy.sparse.csgraph, ordered by first node = networkx's enumeration order).

    on_single_run: the reference's join_ranges raises UnboundLocalError for a component that is ONE run long
    (array_utils.py:659-661); 'raise' keeps that, 'keep' paints the run.
    Returns (pan (D, H, W) uint32 device, per-slice rle_segs or None)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from . import device_tracks as DT
    H, W = (int(v) for v in tiler.image_shape)
    dev = tables[0].r_start.device
    labels, thing_list = list(labels), list(thing_list)
    rank_of = {l: k for k, l in enumerate(labels)}
    out = torch.zeros((n_slices, H, W), dtype=torch.int32, device=dev)
    # ---- node table: every object of every tile, ordered (slice, class, tile, label) = the order in which the
    # reference's per-slice, per-class calls enumerate them (tiles in order, instances ascending: regionprops order)
    cols = []
    for i, t in enumerate(tables):
        if t.n_comp == 0:
            continue
        lab = t.c_label.cpu().numpy()
        cls = lab // label_divisor
        keep = np.isin(cls, labels)
        box = t.c_box.cpu().numpy().astype(np.int64)
        y0, x0 = tiler.yranges[i][0], tiler.xranges[i][0]
        cols.append(np.stack([t.c_slice.cpu().numpy().astype(np.int64), np.vectorize(rank_of.get, otypes=[np.int64])(np.where(keep, cls, labels[0])),
                              np.full(t.n_comp, i, dtype=np.int64), lab, box[:, 0] + y0, box[:, 1] + x0, box[:, 2] + y0,
                              box[:, 3] + x0, t.c_area.cpu().numpy(), np.arange(t.n_comp), keep.astype(np.int64)], axis=1))
    node = np.concatenate(cols) if cols else np.zeros((0, 11), np.int64)
    node = node[node[:, 10] > 0]
    node = node[np.lexsort((node[:, 3], node[:, 2], node[:, 1], node[:, 0]))]
    N = len(node)
    if N == 0:
        return out.view(torch.uint32), ([{l: {} for l in labels} for _ in range(n_slices)] if return_rle else None)
    n_cls = len(labels)
    seg = node[:, 0] * n_cls + node[:, 1]
    tile_of = node[:, 2]
    # ---- lift: runs of every node in the image frame, sorted by (node, start)
    keys, lens = [], []
    for i, t in enumerate(tables):
        if t.n_runs == 0:
            continue
        comp_inst = np.full(max(t.n_comp, 1), -1, dtype=np.int32)
        mine = np.flatnonzero(tile_of == i)
        comp_inst[node[mine, 9]] = mine
        cap = t.n_runs
        key = torch.empty((cap,), dtype=torch.int64, device=dev)
        ln = torch.empty((cap,), dtype=torch.int64, device=dev)
        cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
        work = torch.empty((_hip.query('emp_track_work_elems', t.n_runs),), dtype=torch.int32, device=dev)
        _hip.call('emp_tile_lift', _hip._ptr(t.r_start), _hip._ptr(t.r_len), _hip._ptr(t.r_comp), _hip._ptr(t.c_slice),
                  _hip._ptr(torch.from_numpy(comp_inst).to(dev)), t.n_runs, t.W, W, int(tiler.yranges[i][0]),
                  int(tiler.xranges[i][0]), 0, _hip._ptr(work), _hip._ptr(key), _hip._ptr(ln), _hip._ptr(cnt),
                  _hip.stream())
        keys.append((key, ln, cnt))
    counts = torch.cat([c for _, _, c in keys]).cpu().tolist()
    key = torch.cat([k[:c] for (k, _, _), c in zip(keys, counts)])
    ln = torch.cat([l[:c] for (_, l, _), c in zip(keys, counts)])
    n_runs = int(key.numel())
    key, st, ln, n_runs = DT.sort_runs(key, ln, n_runs)
    off = torch.empty((N + 1,), dtype=torch.int64, device=dev)
    _hip.call('emp_track_offsets', _hip._ptr(key), n_runs, N, _hip._ptr(off), _hip.stream())
    # ---- candidate pairs: boxes of different tiles, same (slice, class), strictly positive intersection
    thing_node = np.isin(np.asarray(labels)[node[:, 1]], thing_list)
    box3 = np.stack([seg, node[:, 4], node[:, 5], seg + 1, node[:, 6], node[:, 7]], axis=1).astype(np.int32)
    pairs = []
    first_of_slice = np.searchsorted(node[:, 0], np.arange(n_slices + 1))
    z = 0
    while z < n_slices:                                   # chunks of slices: the screen is quadratic in the chunk
        z1 = z + 1
        while z1 < n_slices and first_of_slice[z1 + 1] - first_of_slice[z] <= 8192:
            z1 += 1
        a, b = int(first_of_slice[z]), int(first_of_slice[z1])
        if b - a > 1:
            bd = torch.from_numpy(box3[a:b]).to(dev)
            sd = torch.from_numpy(tile_of[a:b].astype(np.int32)).to(dev)
            pr = _hip.box_pairs(bd, src_a=sd, upper_only=True)
            if pr.numel():
                pairs.append(pr.to(torch.int64) + a)
        z = z1
    comp_of = np.arange(N)
    if pairs:
        pr = torch.cat(pairs)
        pr = pr[torch.from_numpy(thing_node).to(dev)[pr[:, 0]]]            # stuff classes are joined unconditionally
        if pr.numel():
            inter = _hip.rle_pair_intersections(st, ln, off, pr.to(torch.int32).contiguous()).cpu().numpy()
            e = pr.cpu().numpy()[inter > 0]                                # object_iou_graph: edge iff iou > 0
            if len(e):
                g = coo_matrix((np.ones(len(e), np.int8), (e[:, 0], e[:, 1])), shape=(N, N))
                _, comp_of = connected_components(g, directed=False)
    # stuff: one object per (slice, class) segment (merge_semantic_from_tiles)
    key_of = np.where(thing_node, comp_of + 0, -1 - seg)
    # clusters in the order of their first node (nodes are sorted by segment, so clusters are grouped by segment and
    # ordered inside it as nx.connected_components enumerates them)
    _, first_node, cl_of = np.unique(key_of, return_index=True, return_inverse=True)
    order = np.argsort(first_node, kind='stable')
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    cl_of = rank[cl_of]                                                    # cluster index of every node
    n_cl = len(order)
    cl_first = first_node[order]
    cl_size = np.bincount(cl_of, minlength=n_cl)
    runs_per_node = np.diff(off.cpu().numpy())
    cl_runs = np.bincount(cl_of, weights=runs_per_node, minlength=n_cl)
    if on_single_run == 'raise' and (cl_runs < 2).any():
        raise UnboundLocalError("local variable 'range2' referenced before assignment")      # _join_ranges :659-661
    cl_thing = thing_node[cl_first]
    dropped = np.zeros(n_cl, dtype=bool)
    if use_overlap and len(tiler.overlap_rle[0]):
        # rle_ioa(overlap, object) > 0.1 for objects seen in one tile only (consensus.py:599-615)
        prefix = _overlap_prefix(tiler, dev)
        hw = H * W
        ov = (prefix[torch.clamp(st + ln, max=hw)] - prefix[torch.clamp(st, max=hw)]).to(torch.int64)
        cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(ov, 0)])
        ov_node = (cs[off[1:]] - cs[off[:-1]]).cpu().numpy()
        single = cl_thing & (cl_size == 1)
        ioa = ov_node[cl_first].astype(np.float64) / node[cl_first, 8].astype(np.float64)
        dropped = single & (ioa > 0.1)
    # new ids: things count up from the smallest label of their (slice, class) call over the surviving clusters;
    # a stuff class keeps the label of its first object
    cl_seg = seg[cl_first]
    seg_min = np.full(int(seg.max()) + 1, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(seg_min, seg, node[:, 3])
    kept = ~dropped
    kept_before = np.cumsum(kept) - kept
    seg_start = np.searchsorted(cl_seg, cl_seg, side='left')               # first cluster of the same segment
    new_id = np.where(cl_thing, seg_min[cl_seg] + kept_before - kept_before[seg_start], node[cl_first, 3])
    new_id = np.where(kept, new_id, 0)
    # ---- paint: every node's runs with its cluster's id, later clusters over earlier ones (fill order of
    # rle_seg_to_pan_seg: classes in `labels` order, instances in dict order); runs are cut at the slice's end
    node_cl = torch.from_numpy(cl_of.astype(np.int32)).to(dev)
    run_cl = torch.empty((n_runs,), dtype=torch.int32, device=dev)
    _hip.call('emp_track_expand', _hip._ptr(off), _hip._ptr(node_cl), N, n_runs, _hip._ptr(run_cl), _hip.stream())
    node_z = torch.from_numpy(node[:, 0].astype(np.int32)).to(dev)
    run_z = torch.empty((n_runs,), dtype=torch.int32, device=dev)
    _hip.call('emp_track_expand', _hip._ptr(off), _hip._ptr(node_z), N, n_runs, _hip._ptr(run_z), _hip.stream())
    hw = H * W
    ln_cut = torch.minimum(ln, torch.clamp(hw - st, min=0))
    _hip.fill_runs_u32(out.view(torch.uint32).reshape(-1), st + run_z.to(torch.int64) * hw, ln_cut, run_cl,
                       _hip.np_to_dev_u32(new_id))
    rles = None
    if return_rle:
        rles = _cluster_rle_segs(st, ln, run_cl, n_cl, kept, new_id, cl_of, cl_seg, node, labels, n_slices)
    return out.view(torch.uint32), rles


def _cluster_rle_segs(st, ln, run_cl, n_cl, kept, new_id, cl_of, cl_seg, node, labels, n_slices):
    """the stitched rle_segs as the reference's merge functions return them (join_ranges of every surviving cluster,
    merged boxes), for callers and tests that want the dict form"""
    from ..array_utils import merge_boxes
    rng, off = _hip.vote_ranges(st, (st + ln).contiguous(), run_cl, n_cl, 1)
    off = off.cpu().numpy().astype(np.int64)
    rng = rng[:int(off[-1])].cpu().numpy()
    n_cls = len(labels)
    segs = [{l: {} for l in labels} for _ in range(n_slices)]
    boxes = {}
    for i in range(len(node)):
        c = int(cl_of[i])
        b = tuple(int(v) for v in node[i, 4:8])
        boxes[c] = b if c not in boxes else tuple(int(v) for v in merge_boxes(boxes[c], b))
    for c in range(n_cl):
        if not kept[c]:
            continue
        z, l = int(cl_seg[c]) // n_cls, labels[int(cl_seg[c]) % n_cls]
        r = rng[off[c]:off[c + 1]]
        segs[z][l][int(new_id[c])] = {'box': boxes[c], 'starts': r[:, 0].copy(), 'runs': r[:, 1] - r[:, 0]}
    return segs


def tiled_panoptic_stack(tile_heads, n_slices, tiler, labels, *, thing_list, label_divisor=1000, use_overlap=True,
                         return_rle=False, on_single_run='raise', **engine_kwargs):
    """Panoptic labels of D slices of a tiled plane.

    tile_heads(i) -> {'sem' (D, C, th, tw) probabilities, 'ctr_hmp' (D, 1, h, w), 'offsets' (D, 2, h, w)} of tile i
    over all D slices (the model forward on the tile's crops, or crops of resident head tensors).
    engine_kwargs: panoptic_stack's (stuff_area, void_label, nms_threshold, nms_kernel, confidence_thr,
    median_kernel_size, coarse_boundaries).
    Returns pan (D, H, W) uint32 on the device (and the per-slice stitched rle_segs if return_rle)."""
    _hip.require_gpu()
    import time
    H, W = tiler.image_shape
    labels, thing_list = list(labels), list(thing_list)
    tables = []
    t_start = time.perf_counter()
    for i in range(len(tiler)):
        h = tile_heads(i)
        pan, emitted = panoptic_stack(h['sem'], h['ctr_hmp'], h['offsets'], thing_list=thing_list,
                                      label_divisor=label_divisor, **engine_kwargs)
        assert len(emitted) == n_slices, "stack shorter than the median kernel"
        th, tw = tiler.yranges[i][1] - tiler.yranges[i][0], tiler.xranges[i][1] - tiler.xranges[i][0]
        # objects = (slice, value) groups: pan_seg_to_rle_seg(force_connected=False), tests/test_tiling.py:36-38
        tables.append(_hip.extract_runs(pan[:, :th, :tw].contiguous(), label_divisor, []))
    t_tiles = time.perf_counter()
    out, stitched = stitch_stack(tables, tiler, n_slices, labels, thing_list, label_divisor, use_overlap,
                                 on_single_run, return_rle)
    torch.cuda.current_stream().synchronize()
    t_end = time.perf_counter()
    TIMERS['tiles_pixels_and_runs'] = TIMERS.get('tiles_pixels_and_runs', 0.0) + t_tiles - t_start
    TIMERS['stitch_and_paint'] = TIMERS.get('stitch_and_paint', 0.0) + t_end - t_tiles
    return (out, stitched) if return_rle else out
