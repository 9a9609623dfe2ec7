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


@pytest.mark.parametrize('split', [(7, 7), (5, 9)])
def test_two_rank_chain_equals_single_rank(split):
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
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bounds, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(np.concatenate([got[0], got[1]]), expected)


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
class _FakeTable:
    """CPU twin of _hip.RunTable (torch tensors on the host) for the numpy assembly code."""


def cpu_run_table(pan_shape, comp_maps, host):
    D, H, W = pan_shape
    rs, rl, rc = [], [], []
    for d in range(D):
        cm = comp_maps[d].reshape(H, W)
        for y in range(H):
            row = cm[y]
            x = 0
            while x < W:
                if row[x] >= 0:
                    x1 = x
                    while x1 < W and row[x1] == row[x]:
                        x1 += 1
                    rs.append(y * W + x); rl.append(x1 - x); rc.append(int(row[x]))
                    x = x1
                else:
                    x += 1
    t = _FakeTable()
    t.D, t.H, t.W = D, H, W
    t.r_start = torch.tensor(rs, dtype=torch.int32)
    t.r_len = torch.tensor(rl, dtype=torch.int32)
    t.r_comp = torch.tensor(rc, dtype=torch.int32)
    t.c_slice = torch.from_numpy(host['c_slice'].astype(np.int32))
    t.n_runs, t.n_comp = len(rs), len(host['c_slice'])
    return t


def _worker_plane(rank, world, port, bounds, axis_name, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        pan = make_stack(seed=5)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        ext = pan[lo:hi + 1] if rank + 1 < world else pan[lo:hi]

        def fake_tables(pan_local, labels, thing_list, label_divisor, group=None):
            host, maps = cpu_tables(ext, list(labels), list(thing_list))
            return cpu_run_table(ext.shape, maps, host), host
        sharded.sharded_tables = fake_tables
        shape3d = pan.shape if axis_name == 'xy' else (pan.shape[1], pan.shape[0], pan.shape[2])
        trs = sharded.sharded_track_plane(torch.zeros((hi - lo, 1, 1)), axis_name, shape3d, lo, [1, 2], [1], DIV)
        q.put((rank, None if trs is None else [{k: (v['box'], v['starts'], v['runs']) for k, v in t.instances.items()}
                                                for t in trs]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('axis_name', ['xy', 'xz'])
def test_two_rank_plane_trackers_equal_single_rank(axis_name):
    """sharded_track_plane over gloo (tables -> rank-0 chain -> partial trackers -> all_gather_object -> stitch)
    against the same functions on one rank.  The device half is replaced by its numpy twin (no GPU here); the yz
    scatter path is covered on the GPU (tests/test_pipeline_gpu.py::test_partial_trackers_stitch...)."""
    from empanada_amd.inference.patterns import _assemble_trackers
    pan = make_stack(seed=5)
    shape3d = pan.shape if axis_name == 'xy' else (pan.shape[1], pan.shape[0], pan.shape[2])
    host, maps = cpu_tables(pan, [1, 2], [1])
    final, first_seen = chain_from_tables(host, pan.shape[0], [1, 2], [1], DIV, 0.25, 0.25)
    table = cpu_run_table(pan.shape, maps, host)
    whole = _assemble_trackers(table, final, host['c_slice'], host['c_cls'], host['c_box'], first_seen, axis_name,
                               shape3d, [1, 2], DIV)
    bounds = np.array([0, 6, pan.shape[0]])
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
    assert got[1] is None
    assert sum(len(t.instances) for t in whole) > 3
    for tr, g in zip(whole, got[0]):
        assert list(tr.instances.keys()) == list(g.keys())
        for k, a in tr.instances.items():
            assert tuple(a['box']) == tuple(g[k][0])
            np.testing.assert_array_equal(a['starts'], g[k][1])
            np.testing.assert_array_equal(a['runs'], g[k][2])


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
