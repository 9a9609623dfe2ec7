"""Throughput of the tiled-plane driver (empanada_amd/inference/tiled.py) at BASELINE configs[4]'s plane shape:
2048 x 2048 planes, C = 5 (background + 3 thing classes + 1 stuff class), 1024-pixel tiles with 128 pixels of overlap
(3 x 3 tiles), D slices; against the untiled whole-plane post-processing of the same heads (planes of this size fit
the MI355X's HBM, so tiling is a compatibility path here, not a necessity).  Post-processing only: the forward is
priced by bench.py.   `PYTHONPATH=. python tools/bench_tiled.py [D]`"""
import sys
import time

import torch

from empanada_amd import synthetic as SY
from empanada_amd.inference import tile, tiled
from empanada_amd.inference.postprocess import panoptic_stack

D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S, TILE, OV, DIV = 2048, 1024, 128, 1000
LABELS, THINGS = [1, 2, 3, 4], [1, 2, 3]
KW = dict(stuff_area=64, void_label=0, nms_threshold=0.1, nms_kernel=7, confidence_thr=0.5, median_kernel_size=3,
          coarse_boundaries=False)


def planted_boxes(seed=7, cell=64, shift=30, p=0.35):
    """z-extruded rectangles, one per 64-pixel cell (with probability p), corners on a 4-pixel grid that also contains
    the tile borders (multiples of 128): a tile never keeps a single row of an object.  The reference's tile merge
    raises on an object that is ONE run long (array_utils.py:659-661, reproduced by this package: DESIGN section 4),
    which random ellipsoids produce at their caps and at tile borders."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lab = np.zeros((D, S, S), dtype=np.uint16)
    classes = [0]
    for cy in range(shift, S - cell, cell):
        for cx in range(shift, S - cell, cell):
            if rng.random() > p:
                continue
            h, w = 4 * rng.integers(3, 11, size=2)
            y0 = cy + 2 + 4 * rng.integers(0, (cell - 4 - h) // 4 + 1)
            x0 = cx + 2 + 4 * rng.integers(0, (cell - 4 - w) // 4 + 1)
            lab[:, y0:y0 + h, x0:x0 + w] = len(classes)
            classes.append(int(rng.integers(1, 5)))
    return lab, np.array(classes, dtype=np.uint8)


def main():
    lab, cls = planted_boxes()
    heads = SY.planted_heads(lab, cls, 'xy', n_classes=4, seed=3, device='cuda')
    tl = tile.Tiler((S, S), TILE, OV)

    def crop(i):
        (y0, y1), (x0, x1) = tl.yranges[i], tl.xranges[i]
        return {k: v[:, :, y0:y1, x0:x1].contiguous() for k, v in heads.items()}

    for rep in range(3):                                  # the last repetition is reported (the first builds caches)
        tiled.TIMERS.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        whole, _ = panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'], thing_list=THINGS, label_divisor=DIV, **KW)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pan = tiled.tiled_panoptic_stack(crop, D, tl, LABELS, thing_list=THINGS, label_divisor=DIV, **KW)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    vox = D * S * S
    n_obj = int(torch.unique(pan.view(torch.int32)).numel()) - 1
    print(f'{D} slices of {S}x{S}, {len(tl)} tiles, {n_obj} labels in the stitched stack')
    print(f'untiled panoptic_stack      : {1e3 * (t1 - t0):9.1f} ms  {vox / (t1 - t0) / 1e6:9.1f} Mvox/s')
    print(f'tiled_panoptic_stack        : {1e3 * (t2 - t1):9.1f} ms  {vox / (t2 - t1) / 1e6:9.1f} Mvox/s')
    if hasattr(tiled, 'TIMERS'):
        for k, v in tiled.TIMERS.items():
            print(f'    {k:24s}: {1e3 * v:9.1f} ms')


if __name__ == '__main__':
    main()
