"""CPU, world_size 2, gloo: the host half of the slice-sharded multi-GPU path
(empanada_amd/inference/sharded.py: merge of per-rank component tables with a one-slice halo, the global
label-propagation chain on rank 0, the broadcast of final labels) gives exactly the labels of the
single-rank chain -- which tests/test_pipeline_gpu.py pins to the reference."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from empanada_amd import synthetic as SY
from empanada_amd.inference import sharded
from empanada_amd.inference.patterns import chain_from_tables
from oracle import rle_seg as OS

DIV = 1000


def cpu_tables(pan, labels, thing_list):
    """numpy twin of patterns.tables_from_stack (which needs the GPU), built on the oracle's CC."""
    D, H, W = pan.shape
    c_slice, c_label, c_area, c_box, c_cls = [], [], [], [], []
    comp_maps = np.full((D, H * W), -1, dtype=np.int64)
    for d in range(D):
        firsts = []
        for l in labels:
            seg = np.where((pan[d] >= l * DIV) & (pan[d] < (l + 1) * DIV), pan[d], 0)
            if l in thing_list:
                cc = OS.connected_components(seg)
                ids = [(k, l * DIV + k) for k in range(1, int(cc.max()) + 1)]
                lab_img = cc
            else:
                vals = [int(v) for v in np.unique(seg) if v]
                ids = [(v, v) for v in vals]
                lab_img = seg
            for k, lab in ids:
                m = lab_img == k
                flat = np.flatnonzero(m.ravel())
                ys, xs = np.nonzero(m)
                firsts.append((int(flat[0]), l, lab, int(m.sum()), (ys.min(), xs.min(), ys.max() + 1, xs.max() + 1), flat))
        firsts.sort(key=lambda t: t[0])                # component order of the run table: first pixel, raster
        for _, l, lab, area, box, flat in firsts:
            comp_maps[d, flat] = len(c_slice)
            c_slice.append(d); c_label.append(lab); c_area.append(area); c_box.append(box); c_cls.append(l)
    c_cls = np.array(c_cls, dtype=np.int64)
    trip = {}
    for d in range(D - 1):
        a, b = comp_maps[d], comp_maps[d + 1]
        m = (a >= 0) & (b >= 0)
        m[m] &= c_cls[a[m]] == c_cls[b[m]]
        for x, y in zip(a[m], b[m]):
            trip[(int(x), int(y))] = trip.get((int(x), int(y)), 0) + 1
    trip = np.array([(a, b, n) for (a, b), n in trip.items()], dtype=np.int64).reshape(-1, 3)
    host = dict(c_slice=np.array(c_slice, dtype=np.int64), c_label=np.array(c_label, dtype=np.int64),
                c_area=np.array(c_area, dtype=np.int64), c_box=np.array(c_box, dtype=np.int32).reshape(-1, 4),
                c_cls=c_cls, trip=trip)
    return host, comp_maps


def make_stack(seed=0, shape=(14, 40, 48)):
    lab, cls = SY.planted_labels(shape, fill=0.25, rmin=4, rmax=9, seed=seed, n_classes=2)
    pan = np.zeros(shape, dtype=np.int64)
    rng = np.random.default_rng(seed)
    for i in range(1, len(cls)):
        # slice-dependent ids so that matching has real work; class 2 = stuff (id 0)
        pan[lab == i] = 1 * DIV + 1 + (i * 7) % 50 if cls[i] == 1 else 2 * DIV
    pan[:, ::9, :] = 0                                  # split objects into several components
    return pan


def paint(comp_maps, final):
    out = np.zeros(comp_maps.shape, dtype=np.int64)
    m = comp_maps >= 0
    out[m] = final[comp_maps[m]]
    return out


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bounds, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pan = make_stack()
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        ext = pan[lo:hi + 1] if rank + 1 < world else pan[lo:hi]
        host, maps = cpu_tables(ext, [1, 2], [1])
        final = sharded.gather_tables_and_chain(host, hi - lo, [1, 2], [1], DIV, 0.25, 0.25, min_size=60, min_span=3)
        q.put((rank, paint(maps[:hi - lo], final)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('split', [(7, 7), (5, 9), (2, 2, 2, 2, 1, 2, 1, 2)])
def test_sharded_chain_equals_single_rank(split):
    """world 2 and world 8 (the node's size, unequal blocks down to a single slice per rank)"""
    pan = make_stack()
    host, maps = cpu_tables(pan, [1, 2], [1])
    final, _ = chain_from_tables(host, pan.shape[0], [1, 2], [1], DIV, 0.25, 0.25)
    final = sharded.filter_labels(host, final, min_size=60, min_span=3)
    expected = paint(maps, final)
    assert len(np.unique(expected)) > 4, "the synthetic stack should keep several instances"
    assert (expected > 0).sum() < (pan > 0).sum(), "the filters should remove something"

    bounds = np.concatenate([[0], np.cumsum(split)])
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    world = len(split)
    procs = [ctx.Process(target=_worker, args=(r, world, port, bounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(np.concatenate([got[r] for r in range(world)]), expected)


def test_shard_bounds():
    np.testing.assert_array_equal(sharded.shard_bounds(10, 4), [0, 3, 6, 8, 10])
    np.testing.assert_array_equal(sharded.shard_bounds(8, 8), np.arange(9))


def test_filter_labels_matches_tracker_filters():
    """filters evaluated on tables == remove_small_objects / remove_pancakes on assembled trackers (oracle)."""
    pan = make_stack(seed=3)
    host, maps = cpu_tables(pan, [1, 2], [1])
    final, _ = chain_from_tables(host, pan.shape[0], [1, 2], [1], DIV, 0.25, 0.25)
    kept = sharded.filter_labels(host, final, min_size=80, min_span=4)
    vol = paint(maps, final).reshape(pan.shape)
    exp = vol.copy()
    for lab in np.unique(vol):
        if lab == 0:
            continue
        zz, yy, xx = np.nonzero(vol == lab)
        spans = [zz.max() - zz.min() + 1, yy.max() - yy.min() + 1, xx.max() - xx.min() + 1]
        if len(zz) < 80 or min(spans) < 4:
            exp[vol == lab] = 0
    np.testing.assert_array_equal(paint(maps, kept).reshape(pan.shape), exp)


def test_host_chain_against_reference_goldens():
    """CPU: chain_from_tables (the host half of track_stack) on the reference's own per-slice panoptic
    stacks reproduces the reference's labels after forward + backward matching (tests/golden/pipeline.npz)."""
    from conftest import load_golden
    g = load_golden('pipeline')
    global DIV
    for i in range(int(g['n'])):
        C = int(g[f'p{i}_par'][0])
        thing = [1] if C == 1 else list(range(1, C))
        labels = [1] if C == 1 else list(range(1, C + 1))
        for name in ('xy', 'xz', 'yz'):
            pans = g[f'p{i}_{name}_pan'].astype(np.int64)
            host, maps = cpu_tables(pans, labels, thing)
            final, _ = chain_from_tables(host, pans.shape[0], labels, thing, DIV, 0.25, 0.25)
            np.testing.assert_array_equal(paint(maps, final).reshape(pans.shape), g[f'p{i}_{name}_bwd'],
                                          err_msg=f'{i} {name}')


# ------------------------------------------------------------------------------------------------ orthoplane
def _worker_plane(rank, world, port, bounds, axis_name, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from empanada_amd.inference import device_tracks as DT
        pan = make_stack(seed=5)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        ext = pan[lo:hi + 1] if rank + 1 < world else pan[lo:hi]
        host, maps = cpu_tables(ext, [1, 2], [1])
        merged, final, first_seen, own, slice0 = sharded.chain_over_ranks(host, hi - lo, [1, 2], [1], DIV, 0.25, 0.25)
        lab, cls, area, box, comp_inst = DT.instance_table(merged, final, first_seen, axis_name, [1, 2])
        mine = np.where(own >= 0, comp_inst[np.maximum(own, 0)], -1)       # instance of each local component
        q.put((rank, dict(final=final, lab=lab, cls=cls, area=area, box=box, slice0=slice0,
                          painted=paint(maps[:hi - lo], np.where(mine >= 0, lab[np.maximum(mine, 0)], 0)),
                          seen={k: list(v.items()) for k, v in first_seen.items()})))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('axis_name', ['xy', 'xz', 'yz'])
@pytest.mark.parametrize('split', [(6, 8), (9, 5)])
def test_two_rank_replicated_chain_and_instance_tables(axis_name, split):
    """chain_over_ranks over gloo (ONE padded all-gather of the packed tables, chain replicated on every rank) and
    the instance table built from it: both ranks hold exactly what a single rank computes over the whole axis --
    labels, dict order, 3D boxes, voxel counts -- and each rank's own components map to the right instances."""
    from empanada_amd.inference import device_tracks as DT
    pan = make_stack(seed=5)
    host, maps = cpu_tables(pan, [1, 2], [1])
    final, first_seen = chain_from_tables(host, pan.shape[0], [1, 2], [1], DIV, 0.25, 0.25)
    lab, cls, area, box, comp_inst = DT.instance_table(host, final, first_seen, axis_name, [1, 2])
    assert len(lab) > 3
    # the table against a direct evaluation on the painted volume (slices along the plane's normal)
    vol = paint(maps, final).reshape(pan.shape)
    k = {'xy': 0, 'xz': 1, 'yz': 2}[axis_name]
    for i in range(len(lab)):
        ss, rr, cc = np.nonzero((vol == lab[i]) & (pan // DIV == cls[i]))
        assert area[i] == len(ss)
        lo3 = [rr.min(), cc.min()]
        hi3 = [rr.max() + 1, cc.max() + 1]
        lo3.insert(k, ss.min())
        hi3.insert(k, ss.max() + 1)
        assert list(box[i]) == lo3 + hi3
    bounds = np.concatenate([[0], np.cumsum(split)])
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_plane, args=(r, 2, port, bounds, axis_name, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        g = got[r]
        np.testing.assert_array_equal(g['final'], final)
        for name, exp in (('lab', lab), ('cls', cls), ('area', area), ('box', box)):
            np.testing.assert_array_equal(g[name], exp)
        assert g['seen'] == {k2: list(v.items()) for k2, v in first_seen.items()}
        assert g['slice0'] == int(bounds[r])
    np.testing.assert_array_equal(np.concatenate([got[0]['painted'], got[1]['painted']]), paint(maps, final))


def _median_twin(prob, ks, thr, want_prob=False):
    """numpy statement of the recursive median + harden (engines.py:68-90, 114-121) with the kernel's interface"""
    x = prob.numpy()
    D = x.shape[0]
    m = (ks - 1) // 2
    out = x.copy()
    for t in range(m, D - m):
        out[t] = np.median(np.concatenate([out[t - m:t], x[t:t + m + 1]]), axis=0)
    sem = torch.from_numpy((out[:, 0] >= thr).astype(np.uint8))
    return (sem, torch.from_numpy(out)) if want_prob else sem


def _worker_median(rank, world, port, bounds, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        x = torch.from_numpy(np.random.default_rng(7).random((int(bounds[-1]), 1, 6, 5)).astype(np.float32))
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        q.put((rank, sharded.median_handover(x[lo:hi], 7, 0.5, median=_median_twin).numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('split', [(7, 7), (3, 11), (4, 5, 5), (3, 3, 3, 5), (3, 5, 4, 3, 6, 3, 4, 4)])
def test_median_handover_equals_whole_axis(split):
    """the rank-to-rank hand-over of filtered history + raw halo reproduces the recursive whole-axis median bit for
    bit, for equal and unequal blocks down to the minimum block size (ks // 2 slices)"""
    bounds = np.concatenate([[0], np.cumsum(split)])
    x = torch.from_numpy(np.random.default_rng(7).random((int(bounds[-1]), 1, 6, 5)).astype(np.float32))
    exp = _median_twin(x, 7, 0.5).numpy()
    world = len(split)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_median, args=(r, world, port, bounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(np.concatenate([got[r] for r in range(world)]), exp)


def _blobby_stack(seed, shape=(40, 48, 56), n=60):
    """random boxes that move, split and merge from slice to slice: overlap matrices with several non-zeros per
    row / column, so the Hungarian callback and the IoA merge rule are exercised"""
    rng = np.random.default_rng(seed)
    D, H, W = shape
    pan = np.zeros(shape, dtype=np.int64)
    for _ in range(n):
        z0, z1 = sorted(rng.integers(0, D, 2))
        y, x = rng.integers(0, H - 8), rng.integers(0, W - 8)
        h, w = rng.integers(3, 12), rng.integers(3, 12)
        cls = 1 if rng.random() < 0.8 else 2
        for z in range(z0, z1 + 1):
            y = int(np.clip(y + rng.integers(-2, 3), 0, H - h))
            x = int(np.clip(x + rng.integers(-2, 3), 0, W - w))
            pan[z, y:y + h, x:x + w] = cls * DIV + 1 + rng.integers(0, 3)
            if rng.random() < 0.2:
                pan[z, y + h // 2, x:x + w] = 0                     # split
    return pan


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
@pytest.mark.parametrize('thr', [(0.25, 0.25), (0.5, 0.1), (0.1, 0.6)])
def test_native_chain_equals_numpy_chain(seed, thr):
    """emp_chain_class (C++) against the numpy statement of the same rules (_ClassChain), incl. the order in which
    labels are first updated"""
    pan = _blobby_stack(seed)
    host, _ = cpu_tables(pan, [1, 2], [1])
    a, fa = chain_from_tables(host, pan.shape[0], [1, 2], [1], DIV, thr[0], thr[1], native=True)
    b, fb = chain_from_tables(host, pan.shape[0], [1, 2], [1], DIV, thr[0], thr[1], native=False)
    np.testing.assert_array_equal(a, b)
    assert {k: list(v.items()) for k, v in fa.items()} == {k: list(v.items()) for k, v in fb.items()}
    assert len(np.unique(a)) > 5


def test_native_chain_error_and_empty_cases():
    host, _ = cpu_tables(np.zeros((5, 8, 8), dtype=np.int64), [1], [1])
    a, fa = chain_from_tables(host, 5, [1], [1], DIV, 0.25, 0.25)
    assert len(a) == 0 and fa == {1: {}}
    pan = np.zeros((3, 8, 8), dtype=np.int64)
    pan[1, 2:5, 2:5] = DIV + 1                                       # empty target slice, then an object
    host, _ = cpu_tables(pan, [1], [1])
    for native in (True, False):
        with pytest.raises(ValueError):
            chain_from_tables(host, 3, [1], [1], DIV, 0.25, 0.0, native=native)


def test_native_lsap_equals_scipy():
    """emp_lsap_maximize (the restatement of the algorithm scipy.optimize.linear_sum_assignment implements) against
    scipy itself: dense, sparse IoU-like, tie-heavy integer, all-zero and rectangular matrices -- the returned
    assignment, not just its value, must be identical, because ties decide which instance keeps a label"""
    from scipy.optimize import linear_sum_assignment
    from empanada_amd import _hip
    lib = _hip.load()
    rng = np.random.default_rng(11)

    def native(m):
        m = np.ascontiguousarray(m, dtype=np.float64)
        k = min(m.shape)
        r, c = np.zeros(k, np.int64), np.zeros(k, np.int64)
        n = lib.emp_lsap_maximize(m.ctypes.data, m.shape[0], m.shape[1], r.ctypes.data, c.ctypes.data)
        return r[:n], c[:n]

    for trial in range(6000):
        nr, nc = (int(v) for v in rng.integers(1, 48, 2))
        kind = trial % 6
        if kind == 0:
            m = rng.random((nr, nc))
        elif kind == 1:
            m = rng.random((nr, nc)) * (rng.random((nr, nc)) < 0.1)
        elif kind == 2:
            m = rng.integers(0, 3, (nr, nc)).astype(float)
        elif kind == 3:
            m = rng.integers(0, 2, (nr, nc)) * 0.5
        elif kind == 4:
            m = np.round(rng.random((nr, nc)), 1) * (rng.random((nr, nc)) < 0.3)
        else:
            m = np.zeros((nr, nc))
        a, b = linear_sum_assignment(m, maximize=True), native(m)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (kind, m.shape)
    for trial in range(40):                                   # the chain's sizes: ~200 components per slice
        nr, nc = (int(v) for v in rng.integers(150, 260, 2))
        m = rng.random((nr, nc)) * (rng.random((nr, nc)) < 0.01)
        if trial % 2:
            m = np.round(m, 1)
        a, b = linear_sum_assignment(m, maximize=True), native(m)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_chain_with_native_lsap_equals_chain_with_scipy(seed):
    pan = _blobby_stack(seed, n=90)
    host, _ = cpu_tables(pan, [1, 2], [1])
    a, fa = chain_from_tables(host, pan.shape[0], [1, 2], [1], DIV, 0.25, 0.25, lsap='native')
    b, fb = chain_from_tables(host, pan.shape[0], [1, 2], [1], DIV, 0.25, 0.25, lsap='scipy')
    np.testing.assert_array_equal(a, b)
    assert {k: list(v.items()) for k, v in fa.items()} == {k: list(v.items()) for k, v in fb.items()}


def _worker_inputs(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import bench
        em, lab, cls = bench.shared_host_inputs((20, 24, 28), rank, 1)
        q.put((rank, em.sum(dtype=np.int64), lab.astype(np.int64).sum(), len(cls), em.flags.writeable))
    finally:
        dist.destroy_process_group()


def test_bench_inputs_are_drawn_once_per_node():
    """bench.shared_host_inputs (N-rank runs): rank 0 draws the synthetic EM and label volumes into /dev/shm, the other
    ranks map them -- every rank ends up with private copies of exactly the arrays a single rank draws, and the scratch
    directory is gone afterwards."""
    import bench
    from empanada_amd import synthetic as SY
    shape = (20, 24, 28)
    lab, cls = SY.planted_labels(shape, fill=0.08, rmin=6, rmax=24, seed=4321, n_classes=1)
    em = SY.em_volume(shape, seed=1234)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_inputs, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict((r, rest) for r, *rest in (q.get(timeout=180) for _ in range(3)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(3):
        assert got[r][0] == em.sum(dtype=np.int64) and got[r][1] == lab.astype(np.int64).sum() and got[r][2] == len(cls)
    base = '/dev/shm' if os.path.isdir('/dev/shm') else __import__('tempfile').gettempdir()
    assert not os.path.exists(os.path.join(base, f'emp_bench_inputs_{port}'))
