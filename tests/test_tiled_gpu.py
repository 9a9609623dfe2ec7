"""GPU: BASELINE configs[4]'s shape at reduced depth -- 2048 x 2048 planes, C = 5 (background + three thing classes +
one stuff class), 1024-pixel tiles with 128 pixels of overlap (3 x 3 tiles), slices split over two virtual ranks.

  * the tiled driver (empanada_amd/inference/tiled.py) against the oracle running the reference's call sequence
    (tests/test_tiling.py:26-47) tile by tile on the same heads: stitched labels identical, ids included;
  * objects that lie inside one tile's exclusive interior keep exactly the pixels the untiled plane gives them;
  * the stitched stack through the slice-sharded stack path with two blocks == one block.
Tile geometry is this repository's (cztile is absent: parity of the geometry itself is unpinned, tests/test_tiles.py
pins the geometry used)."""
import numpy as np
import pytest
import torch

from empanada_amd import synthetic as SY
from oracle import consensus as OC
from oracle import postprocess as OP
from oracle import rle_ops as OR
from oracle import rle_seg as OS

pytestmark = pytest.mark.gpu

D, S, TILE, OV, DIV = 4, 2048, 1024, 128, 1000
LABELS, THINGS = [1, 2, 3, 4], [1, 2, 3]
KW = dict(stuff_area=64, void_label=0, nms_threshold=0.1, nms_kernel=7, confidence_thr=0.5, median_kernel_size=3,
          coarse_boundaries=False)


@pytest.fixture(scope='module')
def plane():
    lab, cls = SY.planted_labels((D, S, S), fill=0.06, rmin=8, rmax=24, seed=55, n_classes=4)
    heads = SY.planted_heads(lab, cls, 'xy', n_classes=4, seed=3, device='cuda')
    return lab, cls, heads


def _oracle_overlap_rle(yranges, xranges, shape):
    """empanada/inference/tile.py:8-52 with the oracle's range functions"""
    y = np.array(OR.rle_voting(np.unique(np.stack(yranges, axis=0), axis=0), 2)).reshape(-1, 2)
    x = np.array(OR.rle_voting(np.unique(np.stack(xranges, axis=0), axis=0), 2)).reshape(-1, 2)
    rs, rr = y[:, 0] * shape[1], (y[:, 1] - y[:, 0]) * shape[1]
    cols = np.concatenate([x + r * shape[1] for r in range(shape[0])], axis=0)
    return OR.merge_rles(rs, rr, cols[:, 0], cols[:, 1] - cols[:, 0])


def test_tiled_stack_equals_oracle_and_shards(plane):
    from empanada_amd.inference import sharded, tile, tiled
    from empanada_amd.inference.postprocess import panoptic_stack
    lab, cls, heads = plane
    tl = tile.Tiler((S, S), TILE, OV)
    assert len(tl) == 9

    def crop(i):
        (y0, y1), (x0, x1) = tl.yranges[i], tl.xranges[i]
        return {k: v[:, :, y0:y1, x0:x1].contiguous() for k, v in heads.items()}

    pan, stitched = tiled.tiled_panoptic_stack(crop, D, tl, LABELS, thing_list=THINGS, label_divisor=DIV,
                                               return_rle=True, **KW)
    got = pan.cpu().numpy()
    assert sum(len(stitched[0][c]) for c in THINGS) > 100 and len(stitched[0][4]) == 1

    # ---- the oracle, tile by tile
    ov = _oracle_overlap_rle(tl.yranges, tl.xranges, (S, S))
    np.testing.assert_array_equal(ov[0], tl.overlap_rle[0])
    np.testing.assert_array_equal(ov[1], tl.overlap_rle[1])
    per_tile = []
    for i in range(len(tl)):
        h = {k: v.cpu().numpy() for k, v in crop(i).items()}
        pans = OP.engine3d_stack([h['sem'][t:t + 1] for t in range(D)], [h['ctr_hmp'][t:t + 1] for t in range(D)],
                                 [h['offsets'][t:t + 1] for t in range(D)], thing_list=THINGS, label_divisor=DIV,
                                 render=True, **KW)
        per_tile.append([OS.pan_seg_to_rle_seg(p.squeeze(), LABELS, DIV, THINGS, force_connected=False) for p in pans])
    for z in range(D):
        exp = np.zeros((S, S), dtype=np.uint32)
        for l in LABELS:
            moved = []
            for i in range(len(tl)):
                (y0, y1), (x0, x1) = tl.yranges[i], tl.xranges[i]
                w = x1 - x0
                insts = {}
                for k, a in per_tile[i][z][l].items():
                    b = a['box']
                    insts[k] = {'box': (b[0] + y0, b[1] + x0, b[2] + y0, b[3] + x0), 'runs': a['runs'],
                                'starts': np.ravel_multi_index((a['starts'] // w + y0, a['starts'] % w + x0), (S, S))}
                moved.append(insts)
            merged = OC.merge_objects_from_tiles(moved, ov) if l in THINGS else OC.merge_semantic_from_tiles(moved)
            assert list(merged.keys()) == list(stitched[z][l].keys())
            OR.numpy_fill_instances(exp.reshape(-1), merged)
        np.testing.assert_array_equal(got[z], exp, err_msg=f'slice {z}')

    # ---- interior objects: same pixels as the untiled plane
    whole, _ = panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'], thing_list=THINGS, label_divisor=DIV, **KW)
    whole = whole.cpu().numpy()
    interior = np.zeros((S, S), dtype=bool)
    interior[:TILE - OV - 64, :TILE - OV - 64] = True           # seen by tile 0 only, away from its borders
    z = D // 2
    ids = np.unique(whole[z][interior])
    ids = [i for i in ids if i and DIV <= i < 4 * DIV and interior[whole[z] == i].all()]
    assert len(ids) > 5
    for i in ids:
        m = whole[z] == i
        vals = np.unique(got[z][m])
        assert len(vals) == 1 and vals[0] // DIV == i // DIV and (got[z] == vals[0]).sum() == m.sum()

    # ---- two virtual ranks over the stitched stack == one rank (slice-sharded stack path)
    single = sharded.sharded_stack_volume(pan, LABELS, THINGS, DIV, 0.25, 0.25, min_size=200, min_span=2)
    from empanada_amd.inference import patterns as PA
    bounds = [0, 1, D]
    tabs, hosts = [], []
    for r in range(2):
        lo, hi = bounds[r], bounds[r + 1]
        ext = pan[lo:hi + 1] if r == 0 else pan[lo:hi]
        t, hh = PA.tables_from_stack(ext.contiguous(), LABELS, THINGS, DIV)
        tabs.append(t); hosts.append(hh)
    merged, own = sharded.merge_rank_tables(hosts, np.array([1, D - 1]))
    final, _ = PA.chain_from_tables(merged, D, LABELS, THINGS, DIV, 0.25, 0.25)
    final = sharded.filter_labels(merged, final, 200, 2)
    slabs = []
    for r in range(2):
        fl = np.zeros(len(own[r]), dtype=np.int64)
        fl[own[r] >= 0] = final[own[r][own[r] >= 0]]
        n_ext = bounds[r + 1] - bounds[r] + (1 if r == 0 else 0)
        slabs.append(sharded.fill_slab(tabs[r], fl, (n_ext, S, S))[:bounds[r + 1] - bounds[r]])
    assert torch.equal(torch.cat(slabs).view(torch.int32), single.view(torch.int32))
    assert len(torch.unique(single.view(torch.int32))) > 50
