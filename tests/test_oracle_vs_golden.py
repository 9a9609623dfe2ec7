"""CPU: the oracle (oracle/) against fixtures produced by the reference itself
(oracle/gen_golden.py) and against the reference's own KATs."""
import numpy as np
import pytest

from conftest import assert_instances_equal, load_golden, unpack_instances, unpack_rle_seg
from oracle import consensus as OC
from oracle import postprocess as OP
from oracle import rle_ops as OR
from oracle import rle_seg as OS


def test_find_centers():
    g = load_golden('find_centers')
    for i in range(int(g['n'])):
        thr, k = g[f'c{i}_par']
        got = OP.find_instance_center(g[f'c{i}_hmp'][None, None], float(thr), int(k))
        np.testing.assert_array_equal(got, g[f'c{i}_ctr'])


def test_group_pixels():
    g = load_golden('group_pixels')
    for i in range(int(g['n'])):
        got = OP.group_pixels(g[f'g{i}_ctr'], g[f'g{i}_off'], step=int(g[f'g{i}_step']))
        np.testing.assert_array_equal(got, g[f'g{i}_ids'], err_msg=f'case {i}')


def test_merge_semantic_and_instance():
    g = load_golden('merge_sem_ins')
    for i in range(int(g['n'])):
        div, stuff, void = (int(x) for x in g[f'm{i}_par'])
        got = OP.merge_semantic_and_instance(g[f'm{i}_sem'], g[f'm{i}_ins'], div,
                                             [int(t) for t in g[f'm{i}_thing']], stuff, void)
        np.testing.assert_array_equal(got, g[f'm{i}_pan'])


def test_median_queue():
    g = load_golden('median_queue')
    for i in range(int(g['n'])):
        xs, ks = g[f'q{i}_x'], int(g[f'q{i}_ks'])
        q = OP.MedianQueue(ks)
        outs, emitted = [], []
        for t in range(len(xs)):
            q.enqueue({'sem': xs[t].copy(), 't': t})
            o = q.get_next(['sem'])
            if o is not None:
                outs.append(o['sem'].copy()); emitted.append(o['t'])
        for o in q.end():
            outs.append(o['sem'].copy()); emitted.append(o['t'])
        np.testing.assert_array_equal(np.array(emitted), g[f'q{i}_emitted'])
        if outs:
            np.testing.assert_array_equal(np.stack(outs), g[f'q{i}_out'])


def test_engines():
    g = load_golden('engines')
    for i in range(int(g['n'])):
        ks, coarse, render, C, nk = (int(x) for x in g[f'e{i}_par'])
        sem, ctr, off = g[f'e{i}_sem'], g[f'e{i}_ctr'], g[f'e{i}_off']
        S, _, H, W = sem.shape
        outs = OP.engine3d_stack(
            [sem[t:t + 1] for t in range(S)], [ctr[t:t + 1] for t in range(S)], [off[t:t + 1] for t in range(S)],
            thing_list=[int(t) for t in g[f'e{i}_thing']], label_divisor=1000, stuff_area=32, void_label=0,
            nms_threshold=0.1, nms_kernel=nk, confidence_thr=float(g[f'e{i}_thr']), median_kernel_size=ks,
            coarse_boundaries=bool(coarse), render=bool(render),
            sizes=[(H - 3, W - 5)] * S if render else None)
        np.testing.assert_array_equal(np.stack(outs), g[f'e{i}_pan'], err_msg=f'engine case {i}')


def test_pan_seg_to_rle_seg():
    g = load_golden('rle_seg')
    for i in range(int(g['n'])):
        pan = g[f'r{i}_pan']
        got = OS.pan_seg_to_rle_seg(pan, [1, 2, 3], 1000, [1, 2], bool(g[f'r{i}_fc']))
        exp = unpack_rle_seg(g, f'r{i}')
        for c in (1, 2, 3):
            assert_instances_equal(got[c], exp.get(c, {}))
        np.testing.assert_array_equal(OS.rle_seg_to_pan_seg(got, pan.shape), g[f'r{i}_back'])


def test_array_utils():
    g = load_golden('array_utils')
    for i in range(6):
        a, b, c = g[f'u{i}_a'], g[f'u{i}_b'], g[f'u{i}_c']
        (sa, ra), (sb, rb), (sc, rc) = OR.rle_encode(a), OR.rle_encode(b), OR.rle_encode(c)
        np.testing.assert_array_equal(sa, g[f'u{i}_sa']); np.testing.assert_array_equal(ra, g[f'u{i}_ra'])
        np.testing.assert_array_equal(OR.rle_decode(sa, ra), a)
        s2, r2 = OR.string_to_rle(OR.rle_to_string(sa, ra))
        np.testing.assert_array_equal(s2, sa); np.testing.assert_array_equal(r2, ra)
        assert OR.rle_intersection(sa, ra, sb, rb) == int(g[f'u{i}_inter']) == len(np.intersect1d(a, b))
        assert OR.rle_iou(sa, ra, sb, rb) == float(g[f'u{i}_iou'])
        assert OR.rle_ioa(sa, ra, sb, rb) == float(g[f'u{i}_ioa'])
        rngs = [np.stack([s, s + r], 1) for s, r in ((sa, ra), (sb, rb), (sc, rc))]
        np.testing.assert_array_equal(OR.vote_by_ranges([r.copy() for r in rngs], 2), g[f'u{i}_vote2'])
        np.testing.assert_array_equal(OR.vote_by_ranges([r.copy() for r in rngs], 3), g[f'u{i}_vote3'])
        np.testing.assert_array_equal(OR.vote_by_ranges([r.copy() for r in rngs], 1), g[f'u{i}_join'])
        ms, mr = OR.merge_rles(sa, ra, sb, rb)
        np.testing.assert_array_equal(ms, g[f'u{i}_ms']); np.testing.assert_array_equal(mr, g[f'u{i}_mr'])
        # the reference's own property (tests/test_array_utils.py:109-153): votes >= 2 of three index sets
        vals, cnt = np.unique(np.concatenate([a, b, c]), return_counts=True)
        v2 = g[f'u{i}_vote2']
        np.testing.assert_array_equal(OR.rle_decode(v2[:, 0], v2[:, 1] - v2[:, 0]), vals[cnt >= 2])
    for i in range(6, 10):
        assert OR.rle_intersection(g[f'u{i}_sa'], g[f'u{i}_ra'], g[f'u{i}_sb'], g[f'u{i}_rb']) == int(g[f'u{i}_inter'])
    for nd in (2, 3):
        a, b = g[f'box{nd}_a'], g[f'box{nd}_b']
        r, c, iou, inter = OR.box_pairs(a, b)
        dense_iou = np.zeros((len(a), len(b))); dense_iou[r, c] = iou
        dense_int = np.zeros((len(a), len(b))); dense_int[r, c] = inter
        np.testing.assert_array_equal(dense_iou, g[f'box{nd}_iou'])
        np.testing.assert_array_equal(dense_int, g[f'box{nd}_inter'])


def test_reference_box_kats():
    """hand-computed tables of the reference's tests/test_array_utils.py:36-107, re-expressed."""
    b = np.array([[0, 0, 10, 10], [5, 5, 15, 15], [10, 10, 20, 20], [0, 0, 5, 5]])
    np.testing.assert_array_equal(OR.box_area(b), [100, 100, 100, 25])
    d = OR.box_iou_dense(b)
    assert d[0, 1] == 25 / 175 and d[0, 2] == 0 and d[0, 3] == 25 / 100 and d[1, 2] == 25 / 175
    assert OR.merge_boxes((0, 0, 10, 10), (5, 5, 15, 15)) == (0, 0, 15, 15)
    b3 = np.array([[0, 0, 0, 10, 10, 10], [5, 5, 5, 15, 15, 15]])
    assert OR.box_iou_dense(b3)[0, 1] == 125 / 1875
    assert OR.merge_boxes((0, 0, 0, 10, 10, 10), (5, 5, 5, 15, 15, 15)) == (0, 0, 0, 15, 15, 15)


def test_join_single_range_quirk():
    with pytest.raises(UnboundLocalError):
        OR.join_ranges([np.array([[0, 5]])])
    assert len(OR.vote_by_ranges([np.array([[0, 5]]), np.array([[5, 9]]), np.array([[3, 6]])], 2)) == 1
    np.testing.assert_array_equal(
        OR.vote_by_ranges([np.array([[0, 5]]), np.array([[5, 9]]), np.array([[3, 6]])], 2), [[3, 6]])


def test_matcher_kat():
    """the reference's tests/test_matcher.py:54-66 known answer."""
    g = load_golden('matcher_kat')
    m = OS.RLEMatcher(1, 1000, 0.25, 0.25, True)
    t = OS.pan_seg_to_rle_seg(g['target'], [1], 1000, [1], False)
    r = OS.pan_seg_to_rle_seg(g['match'], [1], 1000, [1], False)
    m.initialize_target(t[1])
    r[1] = m(r[1], update_target=False)
    np.testing.assert_array_equal(OS.rle_seg_to_pan_seg(r, (200, 200)), g['out'])


def test_trackers_and_chunking():
    g = load_golden('trackers')
    vol = g['vol']
    for name, ax in (('xy', 0), ('xz', 1), ('yz', 2)):
        tr = OS.InstanceTracker(1, 1000, vol.shape, axis=name)
        for idx in range(vol.shape[ax]):
            tr.update(OS.pan_seg_to_rle_seg(np.take(vol, idx, axis=ax), [1], 1000, [1], False)[1], idx)
        tr.finish()
        assert_instances_equal(tr.instances, unpack_instances(g, f't_{name}'))
        filled = OR.numpy_fill_instances(np.zeros_like(vol), tr.instances)
        np.testing.assert_array_equal(filled, g[f't_{name}_filled'])
        if name != 'xz':          # xz carries the reference's row-wrap bug (tracker.py:78-82)
            np.testing.assert_array_equal(filled, vol)


def test_consensus_kats():
    """tests/test_consensus.py:129-194 cases, outputs of the reference."""
    g = load_golden('consensus_kat')
    vols = g['vols']
    shape = vols[0].shape
    trs = [OS.InstanceTracker(1, 1000, shape, axis='xy') for _ in range(3)]
    for z in range(shape[0]):
        for v, tr in zip(vols, trs):
            tr.update(OS.pan_seg_to_rle_seg(v[z], [1], 1000, [1], force_connected=False)[1], z)
    for tr in trs:
        tr.finish()
    for j in range(6):
        vote, iou_thr, bypass = g[f'k{j}_par']
        inst = OC.merge_objects_from_trackers(trs, int(vote), float(iou_thr), bool(bypass))
        assert_instances_equal(inst, unpack_instances(g, f'k{j}_inst'))
        vol = OR.numpy_fill_instances(np.zeros(shape, np.uint32), inst).ravel()
        exp = np.repeat(g[f'k{j}_val'], g[f'k{j}_ln'])
        np.testing.assert_array_equal(vol, exp)
    for j in range(2):
        strs = []
        for v in vols:
            tr = OS.InstanceTracker(1, 1000, shape, axis='xy')
            for z in range(shape[0]):
                tr.update(OS.pan_seg_to_rle_seg((v[z] > 0).astype(np.uint32) * 1000, [1], 1000, [], False)[1], z)
            tr.finish()
            strs.append(tr)
        inst = OC.merge_semantic_from_trackers(strs, int(g[f's{j}_vote']))
        assert_instances_equal(inst, unpack_instances(g, f's{j}_inst'))


def _mg_case(g, i):
    """inputs of forward_multigpu fixture i, regenerated from its seeds (oracle/gen_golden_r2.py)"""
    from empanada_amd import synthetic as SY
    C, ks, seed, n_out = (int(x) for x in g[f'c{i}_par'])
    nthing = 1 if C == 1 else C - 1
    lab, cls = SY.planted_labels((9, 56, 64), fill=0.25, rmin=4, rmax=9, seed=seed, n_classes=nthing)
    heads = SY.planted_heads(lab, cls, 'xy', n_classes=nthing, seed=seed)
    labels = [1] if C == 1 else [1, 2]
    return heads, ks, labels, n_out


def test_get_panoptic_segmentation_golden():
    """P6: oracle get_panoptic_segmentation against the reference's outputs (postprocess.py:298-356)"""
    from empanada_amd import synthetic as SY
    g = load_golden('panoptic_seg')
    for i in range(int(g['n'])):
        C, k, seed = (int(x) for x in g[f'c{i}_par'])
        thr = float(g[f'c{i}_thr'])
        lab, cls = SY.planted_labels((3, 72, 88), fill=0.25, rmin=4, rmax=10, seed=seed, n_classes=max(C - 1, 1))
        heads = SY.planted_heads(lab, cls, 'xy', n_classes=1 if C == 1 else C - 1, seed=seed)
        for z in range(3):
            prob = heads['sem'][z:z + 1].numpy()
            sem = OP.harden_seg(prob, 0.5)
            pan, ctr = OP.get_panoptic_segmentation(sem, heads['ctr_hmp'][z:z + 1].numpy(),
                                                    heads['offsets'][z:z + 1].numpy(), [1], 1000, 16, 0, thr, k)
            np.testing.assert_array_equal(pan, g[f'c{i}_z{z}_pan'], err_msg=f'{i} {z}')
            np.testing.assert_array_equal(ctr, g[f'c{i}_z{z}_ctr'], err_msg=f'{i} {z}')
            assert len(np.unique(pan)) > 2


def test_data_post_fixture_of_the_reference():
    """The reference's own post-processing test (tests/test_data_post.py:13-43): its panoptic ground-truth mask ->
    training targets -> get_panoptic_segmentation.  Fixture = mask, targets and the REFERENCE's output
    (oracle/gen_golden_r3.py).  The oracle must return the same labels and centres, and -- the reference test's own
    assertion -- PQ per class against the mask is 1 to three decimals."""
    from empanada_amd.evaluation import volume_pq
    g = load_golden('data_post')
    sem = g['sem'].astype(np.int64)[None, None]
    pan, ctr = OP.get_panoptic_segmentation(sem, g['ctr_hmp'][None], g['offsets'][None], [2], 1000, 0, 0, 0.1, 7)
    np.testing.assert_array_equal(pan, g['pan'])
    np.testing.assert_array_equal(ctr, g['ctr'])
    assert ctr.shape == (1, 7, 2)
    for c in (1, 2, 3):
        gt = np.where(g['mask'] // 1000 == c, g['mask'], 0)
        pr = np.where(g['pan'][0, 0] // 1000 == c, g['pan'][0, 0], 0)
        pq, n_gt, n_pr, n_m = volume_pq(gt, pr)
        assert n_gt == n_pr == n_m == (7 if c == 2 else 1)
        np.testing.assert_almost_equal(pq, 1.0, decimal=3)


def test_forward_multigpu_golden():
    """forward_multigpu (patterns.py:279-350): the oracle's restatement against the rle_stack the reference sent"""
    g = load_golden('forward_multigpu')
    for i in range(int(g['n'])):
        heads, ks, labels, n_out = _mg_case(g, i)
        items = []
        for z in range(heads['sem'].shape[0]):
            cells = OP.get_instance_cells(heads['ctr_hmp'][z:z + 1].numpy(), heads['offsets'][z:z + 1].numpy(), 0.1, 7,
                                          False)
            items.append((heads['sem'][z:z + 1].numpy(), cells))
        matchers = OS.create_matchers([1], 1000, 0.25, 0.25)
        stack = OS.forward_multigpu(items, matchers, 0.5, ks, labels, 1000, [1], 16, 0)
        assert len(stack) == n_out
        for z, rs in enumerate(stack):
            exp = unpack_rle_seg(g, f'c{i}_z{z}')
            for c in labels:
                assert_instances_equal(rs[c], exp.get(c, {}))
