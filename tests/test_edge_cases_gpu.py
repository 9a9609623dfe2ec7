"""GPU: edge cases of the whole-stack path against the oracle chain -- empty stacks, no centres, everything
foreground, odd sizes (scalar kernel paths), single slice, ks = 1, centre-capacity growth, multi-class with stuff."""
import numpy as np
import pytest
import torch

from empanada_amd import synthetic as SY

pytestmark = pytest.mark.gpu

KW = dict(thing_list=[1], label_divisor=1000, stuff_area=16, void_label=0, nms_threshold=0.1, nms_kernel=7,
          confidence_thr=0.5, median_kernel_size=3)


def oracle_volume(sem, ctr, off, kw, labels, shape, coarse=False, min_size=None, min_span=None):
    from oracle import postprocess as OP
    from oracle import rle_ops as OR
    from oracle import rle_seg as OS
    S = len(sem)
    pans = OP.engine3d_stack([sem[t:t + 1] for t in range(S)], [ctr[t:t + 1] for t in range(S)],
                             [off[t:t + 1] for t in range(S)], coarse_boundaries=coarse, render=True, **kw)
    pans = [p.squeeze(0) if p.ndim == 3 else p for p in pans]
    pans = [p.reshape(shape[1:]) for p in pans]
    matchers = OS.create_matchers(kw['thing_list'], kw['label_divisor'], 0.25, 0.25)
    stack = OS.forward_matching(pans, matchers, labels, kw['label_divisor'], kw['thing_list'])
    trs = OS.create_axis_trackers(['xy'], labels, kw['label_divisor'], (len(pans),) + shape[1:])['xy']
    for idx, rs in OS.backward_matching(stack, matchers, len(pans)):
        OS.update_trackers(rs, idx, trs)
    OS.finish_tracking(trs)
    vol = np.zeros((len(pans),) + shape[1:], np.uint32)
    for tr in trs:
        if min_size is not None:
            OS.remove_small_objects(tr, min_size)
        if min_span is not None:
            OS.remove_pancakes(tr, min_span)
        OR.numpy_fill_instances(vol, tr.instances)
    return np.stack(pans), vol


def product_volume(sem, ctr, off, kw, labels, coarse=False, min_size=None, min_span=None):
    from empanada_amd.inference import sharded
    from empanada_amd.inference.postprocess import panoptic_stack
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pan, emitted = panoptic_stack(t(sem), t(ctr), t(off), coarse_boundaries=coarse, **kw)
    vol = sharded.sharded_stack_volume(pan, labels, kw['thing_list'], kw['label_divisor'], 0.25, 0.25,
                                       min_size=min_size, min_span=min_span)
    return pan.cpu().numpy().astype(np.int64), vol.cpu().numpy()


def check(sem, ctr, off, kw=KW, labels=(1,), coarse=False, **flt):
    shape = (sem.shape[0],) + sem.shape[-2:]
    epan, evol = oracle_volume(sem, ctr, off, kw, list(labels), shape, coarse, **flt)
    gpan, gvol = product_volume(sem, ctr, off, kw, list(labels), coarse, **flt)
    np.testing.assert_array_equal(gpan, epan)
    np.testing.assert_array_equal(gvol, evol)
    return gvol


def test_all_background_and_no_centres():
    sem = np.full((5, 1, 32, 48), 0.1, np.float32)
    ctr = np.zeros((5, 1, 32, 48), np.float32)
    off = np.zeros((5, 2, 32, 48), np.float32)
    assert check(sem, ctr, off).max() == 0
    sem[:] = 0.9            # foreground everywhere but no centre: thing pixels with instance 0 stay void
    assert check(sem, ctr, off).max() == 0


def test_everything_foreground_single_centre():
    sem = np.full((4, 1, 40, 40), 0.9, np.float32)
    ctr = np.zeros((4, 1, 40, 40), np.float32)
    ctr[:, 0, 20, 20] = 1.0
    off = np.zeros((4, 2, 40, 40), np.float32)
    vol = check(sem, ctr, off)
    assert (vol == 1001).all()


@pytest.mark.parametrize('shape', [(6, 37, 53), (1, 64, 64), (3, 33, 130)])
@pytest.mark.parametrize('ks', [1, 3])
def test_odd_sizes_and_short_stacks(shape, ks):
    D, H, W = shape
    lab, cls = SY.planted_labels(shape, fill=0.25, rmin=3, rmax=9, seed=D * 7 + H)
    h = SY.planted_heads(lab, cls, 'xy', seed=H)
    kw = dict(KW, median_kernel_size=ks)
    check(h['sem'].numpy(), h['ctr_hmp'].numpy(), h['offsets'].numpy(), kw, min_size=20, min_span=2)


def test_centre_capacity_grows():
    """more than 256 centres in a slice: the first capacity overflows and the host retries with a larger one"""
    rng = np.random.default_rng(0)
    H = W = 256
    ctr = np.zeros((2, 1, H, W), np.float32)
    ys, xs = np.meshgrid(np.arange(4, H, 12), np.arange(4, W, 12), indexing='ij')
    ctr[:, 0, ys.ravel(), xs.ravel()] = (0.5 + 0.5 * rng.random(ys.size)).astype(np.float32)
    assert ys.size > 256
    sem = (rng.random((2, 1, H, W)) > 0.3).astype(np.float32)
    off = rng.normal(0, 2, (2, 2, H, W)).astype(np.float32)
    vol = check(sem, ctr, off)
    assert len(np.unique(vol)) > 100


def test_too_many_centres_raises():
    from empanada_amd import _hip
    from empanada_amd.inference.postprocess import centers_batched
    hm = torch.zeros((1, 1, 300, 300), device='cuda')
    hm[0, 0, ::2, ::2] = torch.rand(150, 150, device='cuda') * 0.5 + 0.5       # 22500 isolated maxima at k = 1
    with pytest.raises(_hip.HipError):
        centers_batched(hm, 0.1, 1)


def test_multiclass_with_stuff_and_coarse_heads():
    shape = (7, 64, 64)
    lab, cls = SY.planted_labels(shape, fill=0.3, rmin=4, rmax=10, seed=21, n_classes=3)
    h = SY.planted_heads(lab, cls, 'xy', n_classes=3, seed=5, coarse=True)
    kw = dict(KW, thing_list=[1, 2], stuff_area=40)
    vol = check(h['sem'].numpy(), h['ctr_hmp'].numpy(), h['offsets'].numpy(), kw, labels=(1, 2, 3), coarse=True)
    assert (vol == 3000).any() and (vol // 1000 == 1).any() and (vol // 1000 == 2).any()


def test_argument_errors_are_loud():
    from empanada_amd import _hip
    from empanada_amd.inference.postprocess import group_pixels, panoptic_stack
    with pytest.raises(ValueError):
        group_pixels(torch.zeros(1, 2, dtype=torch.long), torch.zeros(2, 2, 8, 8))
    with pytest.raises(_hip.HipError):     # a queue item of another shape would be read out of bounds by the kernel
        _hip.median_step([torch.rand(1, 1, 8, 8).cuda(), torch.rand(0, 1, 8, 8).cuda(), torch.rand(1, 1, 8, 8).cuda()])
    with pytest.raises(_hip.HipError):
        _hip.median_step([torch.rand(1, 1, 8, 8).cuda().double()] * 3)
    with pytest.raises(_hip.HipError):     # ids at another resolution than the class map / up
        _hip.fuse_panoptic(torch.zeros(1, 8, 8, dtype=torch.uint8).cuda(),
                           torch.zeros(1, 4, 4, dtype=torch.int16).cuda(), 4, 2, [1], 1000, 16, 0, up=1)
    with pytest.raises(_hip.HipError):     # one centre list per slice
        _hip.group_pixels(torch.zeros(1, 4, dtype=torch.int32).cuda(), torch.zeros(2, dtype=torch.int32).cuda(),
                          torch.zeros(2, 2, 8, 8).cuda(), 1)
    with pytest.raises(_hip.HipError):     # heads at a resolution the step does not explain
        panoptic_stack(torch.rand(2, 1, 30, 30).cuda(), torch.rand(2, 1, 30, 30).cuda(), torch.rand(2, 2, 30, 30).cuda(),
                       coarse_boundaries=True, **KW)


# ------------------------------------------------------------------------------------------------ orthoplane edges
def _ortho_oracle(heads_by_axis, shape, kw, min_size, min_span, vote=2, iou=0.75):
    from oracle import consensus as OC
    from oracle import postprocess as OP
    from oracle import rle_ops as OR
    from oracle import rle_seg as OS
    trackers = OS.create_axis_trackers(['xy', 'xz', 'yz'], [1], kw['label_divisor'], shape)
    for axis in ('xy', 'xz', 'yz'):
        sem, ctr, off = heads_by_axis[axis]
        n = sem.shape[0]
        pans = OP.engine3d_stack([sem[t:t + 1] for t in range(n)], [ctr[t:t + 1] for t in range(n)],
                                 [off[t:t + 1] for t in range(n)], coarse_boundaries=False, render=True, **kw)
        pans = [p.squeeze() for p in pans]
        matchers = OS.create_matchers([1], kw['label_divisor'], 0.25, 0.25)
        stack = OS.forward_matching(pans, matchers, [1], kw['label_divisor'], [1])
        for idx, rs in OS.backward_matching(stack, matchers, n):
            OS.update_trackers(rs, idx, trackers[axis])
        OS.finish_tracking(trackers[axis])
        for tr in trackers[axis]:
            OS.remove_small_objects(tr, min_size)
            OS.remove_pancakes(tr, min_span)
    con = OC.create_instance_consensus([t for a in ('xy', 'xz', 'yz') for t in trackers[a]], vote, iou, False)
    OS.remove_small_objects(con, min_size)
    OS.remove_pancakes(con, min_span)
    return OR.numpy_fill_instances(np.zeros(shape, np.uint32), con.instances), len(con.instances)


def _ortho_product(heads_by_axis, shape, kw, min_size, min_span, vote=2, iou=0.75):
    from empanada_amd.inference import sharded
    planes, base = {}, 0
    for axis in ('xy', 'xz', 'yz'):
        sem, ctr, off = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in heads_by_axis[axis])
        pan = sharded.sharded_panoptic_stack(sem, ctr, off, coarse_boundaries=False, **kw)
        planes[axis] = sharded.track_plane(pan, axis, shape, [1], [1], kw['label_divisor'], 0.25, 0.25, inst_base=base)
        base += planes[axis].n_inst
    cons, vols, _ = sharded.consensus_volume(planes, shape, [1], [1], vote, iou, False, min_size, min_span)
    return vols[1].cpu().numpy().astype(np.uint32), int(cons[1].alive.sum()), planes


def _heads_of(lab, cls, drop_plane=None):
    out = {}
    for axis in ('xy', 'xz', 'yz'):
        h = SY.planted_heads(lab, cls, axis, seed=7)
        sem, ctr, off = (h[k].numpy() for k in ('sem', 'ctr_hmp', 'offsets'))
        if axis == drop_plane:                       # this plane sees nothing at all
            sem = np.full_like(sem, 0.05)
            ctr = np.zeros_like(ctr)
            off = np.zeros_like(off)
        out[axis] = (sem, ctr, off)
    return out


def test_orthoplane_empty_volume():
    """no object anywhere: three empty planes, empty run tables, empty consensus, zero volume -- no kernel may trip over
    a zero-length table"""
    shape = (12, 16, 20)
    lab = np.zeros(shape, np.uint16)
    cls = np.zeros(1, np.uint8)
    heads = _heads_of(lab, cls)
    for axis in heads:                               # keep the noise below the threshold everywhere
        heads[axis] = (np.minimum(heads[axis][0], 0.2), heads[axis][1], heads[axis][2])
    got, n, planes = _ortho_product(heads, shape, KW, 10, 2)
    assert n == 0 and got.shape == shape and not got.any()
    assert all(p.n_inst == 0 and p.n_runs == 0 for p in planes.values())
    exp, ne = _ortho_oracle(heads, shape, KW, 10, 2)
    assert ne == 0 and not exp.any()


def test_orthoplane_single_object_and_a_blind_plane():
    """one object: (a) seen by all three planes; (b) one plane blind -- two votes still make the consensus
    (pixel_vote_thr 2 of 3, min_cluster_size 2); (c) two planes blind -- nothing survives.  Bit-identical to the oracle."""
    shape = (24, 28, 32)
    lab = np.zeros(shape, np.uint16)
    zz, yy, xx = np.ogrid[:24, :28, :32]
    lab[((zz - 12) / 7.0) ** 2 + ((yy - 14) / 8.0) ** 2 + ((xx - 15) / 9.0) ** 2 <= 1.0] = 1
    cls = np.array([0, 1], np.uint8)
    for drop, expect in ((None, 1), ('xz', 1)):
        heads = _heads_of(lab, cls, drop)
        got, n, _ = _ortho_product(heads, shape, KW, 50, 3)
        exp, ne = _ortho_oracle(heads, shape, KW, 50, 3)
        assert n == ne == expect, (drop, n, ne)
        np.testing.assert_array_equal(got, exp)
        assert got.max() == 1 and (got > 0).sum() > 0.8 * (lab > 0).sum()
    heads = _heads_of(lab, cls, 'xz')
    blind = _heads_of(lab, cls, 'yz')
    heads['yz'] = blind['yz']
    got, n, _ = _ortho_product(heads, shape, KW, 50, 3)
    exp, ne = _ortho_oracle(heads, shape, KW, 50, 3)
    assert n == ne == 0 and not got.any() and not exp.any()


def test_degenerate_class_does_not_lose_the_volume():
    """A stuff class seen by ONE plane only is voted empty; the reference (and empanada_amd.consensus, bug-compatible)
    raises IndexError at the very end of the run (consensus.py:331-339).  The volume driver keeps the finished classes:
    on_degenerate='zero' (default) warns and writes that class as zeros, 'raise' keeps the crash; the thing class is the
    same volume either way.  pixel_vote_thr == 1 with a one-run object (UnboundLocalError out of join_ranges) likewise."""
    from empanada_amd.inference import sharded
    shape = (16, 20, 24)
    div = 1000
    vol = np.zeros(shape, np.int64)
    vol[3:12, 4:14, 5:17] = 1 * div + 1                      # one thing object, seen by every plane
    stuff = np.zeros(shape, np.int64)
    stuff[2:6, 15:19, 2:10] = 2 * div                        # stuff class 2 (clear of the thing object)

    def planes_of(with_stuff_in):
        planes, base = {}, 0
        for ax, axis in enumerate(('xy', 'xz', 'yz')):
            v = vol + (stuff if axis in with_stuff_in else 0)
            pan = torch.from_numpy(np.ascontiguousarray(np.moveaxis(v, ax, 0)).astype(np.int32)).cuda().view(torch.uint32)
            planes[axis] = sharded.track_plane(pan, axis, shape, [1, 2], [1], div, 0.25, 0.25, inst_base=base)
            base += planes[axis].n_inst
        return planes

    _, ok, _ = sharded.consensus_volume(planes_of(('xy', 'xz', 'yz')), shape, [1, 2], [1], 2, 0.75, False)
    assert ok[2].dtype == torch.uint8 and int(ok[2].sum()) == int((stuff > 0).sum())
    with pytest.warns(RuntimeWarning, match='class 2 is degenerate'):
        cons, vols, _ = sharded.consensus_volume(planes_of(('xy',)), shape, [1, 2], [1], 2, 0.75, False)
    assert not vols[2].any() and cons[2].n == 0
    assert torch.equal(vols[1].view(torch.int32), ok[1].view(torch.int32)) and int((vols[1].view(torch.int32) == 1).sum()) == 9 * 10 * 12
    with pytest.raises(IndexError):
        sharded.consensus_volume(planes_of(('xy',)), shape, [1, 2], [1], 2, 0.75, False, on_degenerate='raise')
