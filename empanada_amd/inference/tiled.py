"""Tiled inference of planes that are cut into overlapping tiles (BASELINE configs[4]: 2048 x 2048 planes, C = 5).

The reference ships the pieces -- ``Tiler`` (empanada/inference/tile.py:54-194), ``merge_objects_from_tiles`` /
``merge_semantic_from_tiles`` (empanada/consensus.py:471-625) -- and its test shows the call sequence
(tests/test_tiling.py:26-47): every tile's panoptic image -> ``pan_seg_to_rle_seg(..., force_connected=False)`` ->
``translate_rle_seg`` -> per class ``merge_objects_from_tiles`` -> one rle_seg for the plane.  This module is that
sequence as a driver over a whole stack of slices:

  * tile stacks go through the whole-stack post-processing one tile position at a time (`panoptic_stack`: the recursive
    median runs along z inside each tile position, all slices of a tile in five kernel groups);
  * run extraction for ALL slices of a tile is one pass of the run kernels (`stack_to_rle_segs`);
  * per slice, the tiles' instances are stitched with the reference's merge (box screening, pair intersections and
    range joins on the GPU) and painted into the plane's label image (emp_fill_runs_u32).
The stitched (D, H, W) stack then enters ``track_stack`` / ``sharded`` like any other plane.
"""
import numpy as np
import torch

from .. import _hip
from ..consensus import merge_objects_from_tiles, merge_semantic_from_tiles
from .postprocess import panoptic_stack
from .rle import stack_to_rle_segs

__all__ = ['stitch_slice', 'tiled_panoptic_stack']

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


def tiled_panoptic_stack(tile_heads, n_slices, tiler, labels, *, thing_list, label_divisor=1000, use_overlap=True,
                         return_rle=False, **engine_kwargs):
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
    per_tile = []
    t_start = time.perf_counter()
    for i in range(len(tiler)):
        h = tile_heads(i)
        pan, emitted = panoptic_stack(h['sem'], h['ctr_hmp'], h['offsets'], thing_list=thing_list,
                                      label_divisor=label_divisor, **engine_kwargs)
        assert len(emitted) == n_slices, "stack shorter than the median kernel"
        th, tw = tiler.yranges[i][1] - tiler.yranges[i][0], tiler.xranges[i][1] - tiler.xranges[i][0]
        segs, _ = stack_to_rle_segs(pan[:, :th, :tw].contiguous(), labels, label_divisor, thing_list,
                                    force_connected=False)
        per_tile.append(segs)
    out = torch.zeros((n_slices, H, W), dtype=torch.int32, device='cuda').view(torch.uint32)
    stitched = []
    t_tiles = time.perf_counter()
    t_paint = 0.0
    for z in range(n_slices):
        rs = stitch_slice([per_tile[i][z] for i in range(len(tiler))], tiler, labels, thing_list, use_overlap)
        tp = time.perf_counter()
        _paint(rs, (H, W), out[z])
        t_paint += time.perf_counter() - tp
        if return_rle:
            stitched.append(rs)
    t_end = time.perf_counter()
    TIMERS['tiles_pixels_and_runs'] = TIMERS.get('tiles_pixels_and_runs', 0.0) + t_tiles - t_start
    TIMERS['stitch'] = TIMERS.get('stitch', 0.0) + t_end - t_tiles - t_paint
    TIMERS['paint'] = TIMERS.get('paint', 0.0) + t_paint
    return (out, stitched) if return_rle else out
