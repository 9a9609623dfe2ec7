"""GPU: each HIP kernel group against the oracle (bit-exact) on seeded inputs and against the
fixtures produced by the reference (tests/golden)."""
import numpy as np
import pytest
import torch

from conftest import assert_instances_equal, load_golden, unpack_rle_seg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from empanada_amd import _hip
    _hip.load()
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return _hip


def _oracle_median_stack(x, ks, thr):
    """oracle/postprocess.MedianQueue over a (D,C,H,W) stack -> filtered probs, emitted order."""
    from oracle import postprocess as OP
    q = OP.MedianQueue(ks)
    outs = []
    for t in range(len(x)):
        q.enqueue({'sem': x[t:t + 1].copy()})
        o = q.get_next(['sem'])
        if o is not None:
            outs.append(o['sem'].copy())
    outs += [o['sem'].copy() for o in q.end()]
    filt = np.concatenate(outs, axis=0)
    sem = np.concatenate([OP.harden_seg(f[None], thr)[0] for f in filt], axis=0)
    return filt, sem


@pytest.mark.parametrize('ks', [1, 3, 5, 7, 9, 11])
@pytest.mark.parametrize('C', [1, 3])
def test_median_harden_stack(hip, ks, C):
    rng = np.random.default_rng(ks * 10 + C)
    D, H, W = max(ks, 13), 24, 40
    x = rng.random((D, C, H, W), dtype=np.float32)
    x[:, :, 0, :8] = 0.5        # exact ties with the threshold / between channels
    filt, sem = _oracle_median_stack(x, ks, 0.5)
    gsem, gfilt = hip.median_harden_stack(torch.from_numpy(x).cuda(), ks, 0.5, want_prob=True)
    np.testing.assert_array_equal(gfilt.cpu().numpy(), filt)
    np.testing.assert_array_equal(gsem.cpu().numpy(), sem.astype(np.uint8))


def test_median_golden(hip):
    g = load_golden('median_queue')
    for i in range(int(g['n'])):
        xs, ks = g[f'q{i}_x'], int(g[f'q{i}_ks'])
        if len(xs) < ks:
            continue        # short stacks are degraded on the host (engine tests)
        x = xs[:, 0]        # (n, C=2... stored as (n,1,2,6,7)) -> (n,2,6,7)
        _, filt = hip.median_harden_stack(torch.from_numpy(x).cuda(), ks, 0.5, want_prob=True)
        np.testing.assert_array_equal(filt.cpu().numpy(), g[f'q{i}_out'][:, 0])


@pytest.mark.parametrize('ks', [3, 7, 11])
def test_median_step(hip, ks):
    rng = np.random.default_rng(ks)
    xs = [rng.random((1, 2, 33, 17), dtype=np.float32) for _ in range(ks)]
    exp = np.sort(np.stack(xs), axis=0)[ks // 2]
    got = hip.median_step([torch.from_numpy(x).cuda() for x in xs])
    np.testing.assert_array_equal(got.cpu().numpy(), exp)


def test_find_centers_golden_and_random(hip):
    from oracle import postprocess as OP
    g = load_golden('find_centers')
    cases = [(g[f'c{i}_hmp'], float(g[f'c{i}_par'][0]), int(g[f'c{i}_par'][1]), g[f'c{i}_ctr'])
             for i in range(int(g['n']))]
    rng = np.random.default_rng(5)
    for (h, w, k) in [(130, 257, 7), (64, 64, 3), (16, 300, 5), (200, 70, 8), (65, 129, 2)]:
        hm = (rng.random((h, w), dtype=np.float32) ** 8).astype(np.float32)
        hm[3:6, 10:14] = 0.7
        cases.append((hm, 0.1, k, OP.find_instance_center(hm[None, None], 0.1, k)))
    for hm, thr, k, exp in cases:
        idx, cnt = hip.find_centers(torch.from_numpy(hm[None]).cuda(), thr, k, cap=4096)
        n = int(cnt[0])
        assert n == len(exp)
        flat = idx[0, :n].cpu().numpy()
        got = np.stack([flat // hm.shape[1], flat % hm.shape[1]], axis=1)
        np.testing.assert_array_equal(got, exp)


def test_find_centers_batched_matches_single(hip):
    rng = np.random.default_rng(6)
    hm = (rng.random((5, 96, 80), dtype=np.float32) ** 6).astype(np.float32)
    hm[2] = 0            # a slice without centres
    idx, cnt = hip.find_centers(torch.from_numpy(hm).cuda(), 0.1, 7, cap=512)
    from oracle import postprocess as OP
    for d in range(5):
        exp = OP.find_instance_center(hm[d][None, None], 0.1, 7)
        assert int(cnt[d]) == len(exp)
        flat = idx[d, :len(exp)].cpu().numpy()
        np.testing.assert_array_equal(np.stack([flat // 80, flat % 80], 1).reshape(-1, 2), exp)


def _group(hip, ctr, off, step):
    h, w = off.shape[-2:]
    K = len(ctr)
    cap = max(K, 1)
    idx = torch.zeros((1, cap), dtype=torch.int32)
    idx[0, :K] = torch.from_numpy((ctr[:, 0] * w + ctr[:, 1]).astype(np.int32))
    cnt = torch.tensor([K], dtype=torch.int32)
    ids = hip.group_pixels(idx.cuda(), cnt.cuda(), torch.from_numpy(off).cuda(), step)
    return ids.cpu().numpy().astype(np.int64)


def test_group_pixels_golden(hip):
    g = load_golden('group_pixels')
    for i in range(int(g['n'])):
        got = _group(hip, g[f'g{i}_ctr'], g[f'g{i}_off'], int(g[f'g{i}_step']))
        np.testing.assert_array_equal(got, g[f'g{i}_ids'], err_msg=f'case {i}')


def test_group_pixels_thing_mask(hip):
    """with a semantic map, non-thing pixels are skipped (id 0); thing pixels are unchanged"""
    from oracle import postprocess as OP
    rng = np.random.default_rng(77)
    h, w, K = 70, 132, 37
    ctr = np.stack([rng.integers(0, h, K), rng.integers(0, w, K)], axis=1).astype(np.int64)
    off = rng.normal(0, 5, (1, 2, h, w)).astype(np.float32)
    sem = np.repeat(np.repeat(rng.integers(0, 3, (h // 10, w // 12)), 10, 0), 12, 1).astype(np.uint8)
    exp = OP.group_pixels(ctr, off, step=1)[0] * np.isin(sem, [1])
    idx = torch.from_numpy((ctr[:, 0] * w + ctr[:, 1]).astype(np.int32))[None].cuda()
    cnt = torch.tensor([K], dtype=torch.int32).cuda()
    ids = hip.group_pixels(idx, cnt, torch.from_numpy(off).cuda(), 1, sem=torch.from_numpy(sem)[None].cuda(),
                           thing_list=[1])
    np.testing.assert_array_equal(ids[0].cpu().numpy().astype(np.int64), exp)


@pytest.mark.parametrize('K,step', [(1, 1), (20, 1), (21, 1), (333, 1), (64, 4), (7, 4)])
def test_group_pixels_vs_oracle(hip, K, step):
    from oracle import postprocess as OP
    rng = np.random.default_rng(K * 7 + step)
    h, w = 96, 160
    ctr = np.stack([rng.integers(0, h, K), rng.integers(0, w, K)], axis=1).astype(np.int64)
    # integer offsets produce many exact distance ties; the fractional part produces near ties
    off = rng.integers(-12, 13, (1, 2, h, w)).astype(np.float32) * step
    off[0, :, : h // 2] += rng.normal(0, 1e-3, (2, h // 2, w)).astype(np.float32)
    np.testing.assert_array_equal(_group(hip, ctr, off, step), OP.group_pixels(ctr, off, step=step))


@pytest.mark.parametrize('K,step,noise', [(9, 1, 0.0), (20, 1, 0.3), (21, 1, 0.0), (150, 1, 0.5), (900, 1, 0.2),
                                           (40, 4, 0.0), (300, 4, 1.0)])
def test_group_pixels_structured_offsets(hip, K, step, noise):
    """offsets that point at nearby centres (what a trained head emits): the per-block candidate pruning is active.
    Integer offsets give exact ties between equidistant centres (first index must win), a band of huge offsets lands
    farther than 1e5 from every centre (id 0 when K > 20), one row holds NaN / inf offsets (full walk)."""
    from oracle import postprocess as OP
    rng = np.random.default_rng(K * 13 + step)
    h, w = 128, 192
    ctr = np.stack([rng.integers(0, h, K), rng.integers(0, w, K)], axis=1).astype(np.int64)
    ctr[K // 2] = ctr[0]                                         # a duplicated centre: the lower index wins
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    d2 = (yy[..., None] - ctr[:, 0]) ** 2 + (xx[..., None] - ctr[:, 1]) ** 2
    near = ctr[np.argmin(d2, axis=-1)]
    off = np.stack([near[..., 0] - yy, near[..., 1] - xx]).astype(np.float32)[None] * step
    off[0, :, :, : w // 2] *= 0.5                                # half way: equidistant pairs of centres appear
    off += rng.normal(0, noise, off.shape).astype(np.float32)
    off[0, :, 40:44] += 3e5                                      # beyond the 1e5 ceiling
    off[0, 0, 50, :20] = np.nan
    off[0, 1, 50, 20:40] = np.inf
    np.testing.assert_array_equal(_group(hip, ctr, off, step), OP.group_pixels(ctr, off, step=step))


def test_fuse_golden_and_random(hip):
    from oracle import postprocess as OP
    g = load_golden('merge_sem_ins')
    cases = []
    for i in range(int(g['n'])):
        div, stuff, void = (int(x) for x in g[f'm{i}_par'])
        cases.append((g[f'm{i}_sem'][0], g[f'm{i}_ins'][0], [int(t) for t in g[f'm{i}_thing']], div, stuff, void,
                      g[f'm{i}_pan'][0]))
    rng = np.random.default_rng(9)
    for (thing, nc, K, W) in [([1], 2, 30, 96), ([1, 2, 4], 5, 200, 96), ([2], 3, 3, 96), ([1, 2], 4, 40, 54),
                              ([1], 2, 700, 200)]:
        H = 64
        sem = np.repeat(np.repeat(rng.integers(0, nc, (H // 4, W // 8 + 1)), 4, 0), 8, 1)[:, :W].astype(np.int64)
        ids = np.repeat(np.repeat(rng.integers(0, K + 1, (H // 8, W // 4 + 1)), 8, 0), 4, 1)[:, :W].astype(np.int64)
        ids[::7, ::5] = rng.integers(0, K + 1, ids[::7, ::5].shape)      # break the 4-pixel uniformity
        ins = ids * np.isin(sem, thing)
        exp = OP.merge_semantic_and_instance(sem[None], ins[None], 1000, thing, 40, 0)[0]
        cases.append((sem, ids, thing, 1000, 40, 0, exp))
    for sem, ids, thing, div, stuff, void, exp in cases:
        K = int(ids.max())
        nc = int(sem.max()) + 1
        for dt in (torch.uint32, torch.int64):
            pan = hip.fuse_panoptic(torch.from_numpy(sem.astype(np.uint8))[None].cuda(),
                                    torch.from_numpy(ids.astype(np.int16))[None].cuda().view(torch.uint16),
                                    max(K, 1), nc, thing, div, stuff, void, up=1, out_dtype=dt)
            np.testing.assert_array_equal(pan[0].cpu().numpy().astype(np.int64), exp)


def test_fuse_coarse_upsample(hip):
    from oracle import postprocess as OP
    rng = np.random.default_rng(10)
    H, W, up = 64, 128, 4
    sem = (rng.random((H, W)) < 0.5).astype(np.int64)
    ids_c = rng.integers(0, 9, (H // up, W // up)).astype(np.int64)
    cells = np.repeat(np.repeat(ids_c, up, 0), up, 1)
    exp = OP.get_panoptic_seg(sem[None], cells[None, None].astype(np.float32), 1000, [1], 64, 0)[0]
    pan = hip.fuse_panoptic(torch.from_numpy(sem.astype(np.uint8))[None].cuda(),
                            torch.from_numpy(ids_c.astype(np.int16))[None].cuda().view(torch.uint16),
                            8, 2, [1], 1000, 64, 0, up=up)
    np.testing.assert_array_equal(pan[0].cpu().numpy().astype(np.int64), exp)


def test_scan(hip):
    rng = np.random.default_rng(11)
    for n in [1, 7, 2048, 2049, 100003, 3000000]:
        x = rng.integers(0, 5, n).astype(np.int32)
        got = hip.exclusive_scan_i32(torch.from_numpy(x).cuda()).cpu().numpy()
        np.testing.assert_array_equal(got, np.concatenate([[0], np.cumsum(x)]))


def test_rle_seg_golden(hip):
    from empanada_amd.inference import rle
    g = load_golden('rle_seg')
    for i in range(int(g['n'])):
        pan = g[f'r{i}_pan']
        got = rle.pan_seg_to_rle_seg(pan, [1, 2, 3], 1000, [1, 2], bool(g[f'r{i}_fc']))
        exp = unpack_rle_seg(g, f'r{i}')
        for c in (1, 2, 3):
            assert_instances_equal(got[c], exp.get(c, {}))
        np.testing.assert_array_equal(rle.rle_seg_to_pan_seg(got, pan.shape), g[f'r{i}_back'])


@pytest.mark.parametrize('shape', [(5, 37, 53), (3, 64, 128), (2, 200, 260), (1, 9, 1024)])
def test_runs_cc_vs_oracle(hip, shape):
    """stack extraction: components, boxes, areas and merged runs against the oracle, slice by slice."""
    from empanada_amd.inference import rle
    from oracle import rle_seg as OS
    rng = np.random.default_rng(sum(shape))
    D, H, W = shape
    base = rng.integers(0, 5, (D, H // 3 + 1, W // 2 + 1))
    pan = np.repeat(np.repeat(base, 3, 1), 2, 2)[:, :H, :W]
    noise = rng.random((D, H, W)) < 0.15
    pan = np.where(noise, rng.integers(0, 5, (D, H, W)), pan)
    cls = np.repeat(np.repeat(rng.integers(1, 4, (D, H // 8 + 1, W // 8 + 1)), 8, 1), 8, 2)[:, :H, :W]
    pan = np.where(pan > 0, cls * 1000 + pan, 0).astype(np.int64)
    pan[:, :, -1] = pan[:, :, 0]            # row-wrapping runs
    segs, table = rle.stack_to_rle_segs(hip.np_to_dev_u32(pan), [1, 2, 3], 1000, [1, 2], True)
    for d in range(D):
        exp = OS.pan_seg_to_rle_seg(pan[d], [1, 2, 3], 1000, [1, 2], True)
        for c in (1, 2, 3):
            assert_instances_equal(segs[d][c], exp[c])
    # areas
    areas = table.c_area.cpu().numpy()
    lab = table.c_label.cpu().numpy()
    sl = table.c_slice.cpu().numpy()
    val = table.r_val.cpu().numpy()[table.c_first.cpu().numpy()]
    for c in range(table.n_comp):
        inst = segs[int(sl[c])][int(val[c] // 1000)][int(lab[c])]
        assert areas[c] == inst['runs'].sum()


def test_connected_components_api(hip):
    from empanada_amd.inference import rle
    from oracle import rle_seg as OS
    rng = np.random.default_rng(3)
    seg = np.repeat(np.repeat(rng.integers(0, 3, (20, 25)), 3, 0), 2, 1).astype(np.int64)
    np.testing.assert_array_equal(rle.connected_components(seg), OS.connected_components(seg))


def test_overlap_next(hip):
    from empanada_amd.inference import rle
    rng = np.random.default_rng(12)
    D, H, W = 6, 48, 64
    pan = np.repeat(np.repeat(rng.integers(0, 4, (D, H // 4, W // 4)), 4, 1), 4, 2).astype(np.int64)
    pan = np.where(pan > 0, 1000 + pan, 0)
    pan[rng.random(pan.shape) < 0.05] = 2000
    segs, table = rle.stack_to_rle_segs(hip.np_to_dev_u32(pan), [1, 2], 1000, [1], True)
    trip = hip.overlap_next(table, 1000).cpu().numpy()
    lab = table.c_label.cpu().numpy()
    sl = table.c_slice.cpu().numpy()
    got = {}
    for a, b, n in trip:
        got[(int(a), int(b))] = got.get((int(a), int(b)), 0) + int(n)
    # dense reference: per-pixel pairs of component maps
    comp_map = np.full((D, H * W), -1, dtype=np.int64)
    rs, rl, rc = (x.cpu().numpy() for x in (table.r_start, table.r_len, table.r_comp))
    for s, l, c in zip(rs, rl, rc):
        comp_map[sl[c], s:s + l] = c
    exp = {}
    for d in range(D - 1):
        a, b = comp_map[d], comp_map[d + 1]
        m = (a >= 0) & (b >= 0)
        m &= (lab[np.where(a >= 0, a, 0)] // 1000) == (lab[np.where(b >= 0, b, 0)] // 1000)
        for x, y in zip(a[m], b[m]):
            exp[(int(x), int(y))] = exp.get((int(x), int(y)), 0) + 1
    assert got == exp


def test_pair_intersections_and_iou(hip):
    from empanada_amd import array_utils as AU
    from oracle import rle_ops as OR
    g = load_golden('array_utils')
    for i in range(6):
        a, b = g[f'u{i}_a'], g[f'u{i}_b']
        sa, ra = OR.rle_encode(a); sb, rb = OR.rle_encode(b)
        assert AU.rle_intersection(sa, ra, sb, rb) == int(g[f'u{i}_inter'])
        assert AU.rle_iou(sa, ra, sb, rb) == float(g[f'u{i}_iou'])
        assert AU.rle_ioa(sa, ra, sb, rb) == float(g[f'u{i}_ioa'])
    for i in range(6, 10):      # malformed rles: the literal sweep
        assert AU.rle_intersection(g[f'u{i}_sa'], g[f'u{i}_ra'], g[f'u{i}_sb'], g[f'u{i}_rb']) == int(g[f'u{i}_inter'])
    rng = np.random.default_rng(13)
    insts = []
    for _ in range(12):
        s = rng.integers(0, 5000, 60); r = rng.integers(1, 40, 60)
        insts.append((s, r))
    pairs = [(i, j) for i in range(12) for j in range(12) if i != j]
    got = AU.rle_pair_intersections([s for s, _ in insts], [r for _, r in insts], pairs)
    exp = [OR.rle_intersection(*insts[i], *insts[j]) for i, j in pairs]
    np.testing.assert_array_equal(got, exp)


def test_vote_and_join(hip):
    from empanada_amd import array_utils as AU
    from oracle import rle_ops as OR
    g = load_golden('array_utils')
    for i in range(6):
        rngs = []
        for key in ('a', 'b', 'c'):
            s, r = OR.rle_encode(g[f'u{i}_{key}'])
            rngs.append(np.stack([s, s + r], 1))
        np.testing.assert_array_equal(AU.vote_by_ranges([r.copy() for r in rngs], 2), g[f'u{i}_vote2'])
        np.testing.assert_array_equal(AU.vote_by_ranges([r.copy() for r in rngs], 3), g[f'u{i}_vote3'])
        np.testing.assert_array_equal(AU.vote_by_ranges([r.copy() for r in rngs], 1), g[f'u{i}_join'])
        ms, mr = AU.merge_rles(g[f'u{i}_sa'], g[f'u{i}_ra'], rngs[1][:, 0], rngs[1][:, 1] - rngs[1][:, 0])
        np.testing.assert_array_equal(ms, g[f'u{i}_ms']); np.testing.assert_array_equal(mr, g[f'u{i}_mr'])
    # overlapping ranges inside one list (xz wrap bug shape), big offsets, many groups at once
    rng = np.random.default_rng(14)
    groups = []
    for _ in range(40):
        lst = []
        for _ in range(rng.integers(1, 5)):
            s = rng.integers(0, 2 ** 34, 30) ; e = s + rng.integers(1, 2000, 30)
            lst.append(np.stack([s, e], 1))
        base = lst[0].copy(); base[:, 1] += 500
        lst.append(base)
        groups.append(lst)
    for thr in (1, 2, 3):
        got = AU.vote_groups(groups, thr)
        for gi, lst in enumerate(groups):
            exp = OR.vote_by_ranges([r.copy() for r in lst], thr) if thr > 1 else OR.join_ranges([r.copy() for r in lst])
            np.testing.assert_array_equal(got[gi].reshape(-1, 2), np.asarray(exp).reshape(-1, 2))
    with pytest.raises(UnboundLocalError):
        AU.join_ranges([np.array([[0, 5]])])
    np.testing.assert_array_equal(
        AU.vote_by_ranges([np.array([[0, 5]]), np.array([[5, 9]]), np.array([[3, 6]])], 2), [[3, 6]])


def test_fill_overlapping_order(hip):
    from empanada_amd import array_utils as AU
    from oracle import rle_ops as OR
    rng = np.random.default_rng(15)
    inst = {}
    for k in [7, 3, 12, 5, 9]:           # dict order is not id order; later entries overwrite
        s = np.sort(rng.integers(0, 4000, 25)); r = rng.integers(1, 120, 25)
        inst[k] = {'starts': s, 'runs': r}
    exp = OR.numpy_fill_instances(np.zeros((10, 20, 25), np.uint32), inst)
    got = AU.numpy_fill_instances(np.zeros((10, 20, 25), np.uint32), inst)
    np.testing.assert_array_equal(got, exp)

def test_numpy_fill_instances_into_every_kind_of_volume():
    """array_utils.py:725-737 through the GPU paint: a fresh volume (nothing uploaded, labels copied straight into the
    caller's memory), a volume that already holds labels (later instances overwrite, the rest stays), other integer
    widths and a non-contiguous view (staged copy); the caller's array is filled IN PLACE"""
    from empanada_amd import array_utils as AU
    from oracle import rle_ops as OR
    rng = np.random.default_rng(3)
    shape = (6, 17, 23)
    n = int(np.prod(shape))
    inst = {}
    for i in range(9):
        starts = np.sort(rng.choice(n - 40, size=12, replace=False)).astype(np.int64)
        inst[1000 + 7 * i] = {'box': (0, 0, 0, 1, 1, 1), 'starts': starts, 'runs': rng.integers(1, 30, size=12).astype(np.int64)}
    for dtype in (np.uint32, np.int32, np.uint16, np.uint8):
        # labels are class * divisor + n >= 1: id 0 never occurs (and means "paint nothing" to emp_fill_runs_u32)
        ids = {k % (249 if dtype == np.uint8 else 60000 if dtype == np.uint16 else 2 ** 31) + 1: v for k, v in inst.items()}
        fresh = np.zeros(shape, dtype)
        out = AU.numpy_fill_instances(fresh, ids)
        exp = OR.numpy_fill_instances(np.zeros(shape, dtype), ids)
        np.testing.assert_array_equal(out, exp)
        np.testing.assert_array_equal(fresh, exp)                # in place
        used = rng.integers(0, 200, size=shape).astype(dtype)    # already labelled: upload, paint over, copy back
        exp2 = OR.numpy_fill_instances(used.copy(), ids)
        AU.numpy_fill_instances(used, ids)
        np.testing.assert_array_equal(used, exp2)
    big = np.zeros((6, 17, 46), np.uint32)
    view = big[:, :, ::2]                                        # not contiguous: reshape(-1) copies, the result is returned
    out = AU.numpy_fill_instances(view, inst)
    np.testing.assert_array_equal(out, OR.numpy_fill_instances(np.zeros(shape, np.uint32), inst))
    with pytest.raises(ValueError):
        AU.numpy_fill_instances(np.zeros(shape, np.int64), inst)



def test_sort(hip):
    rng = np.random.default_rng(16)
    k = rng.integers(0, 2 ** 50, 100000).astype(np.uint64)
    v = np.arange(100000, dtype=np.int32)
    ko, vo = hip.sort_u64_i32(torch.from_numpy(k.view(np.int64)).cuda().view(torch.uint64), torch.from_numpy(v).cuda())
    order = np.argsort(k, kind='stable')
    np.testing.assert_array_equal(vo.cpu().numpy(), v[order])


def test_rle_encode_decode(hip):
    """emp_rle_encode / emp_rle_decode vs the reference's golden index lists and the oracle."""
    from empanada_amd import array_utils as AU
    from oracle import rle_ops as OR
    g = load_golden('array_utils')
    for i in range(6):
        for key in 'abc':
            idx = g[f'u{i}_{key}']
            s, r = AU.rle_encode(idx)
            es, er = OR.rle_encode(idx)
            np.testing.assert_array_equal(s, es); np.testing.assert_array_equal(r, er)
            np.testing.assert_array_equal(AU.rle_decode(s, r), idx)
        np.testing.assert_array_equal(AU.rle_encode(g[f'u{i}_a'])[0], g[f'u{i}_sa'])
        np.testing.assert_array_equal(AU.rle_encode(g[f'u{i}_a'])[1], g[f'u{i}_ra'])
    rng = np.random.default_rng(5)
    for n, p in ((1, 1.0), (2, 0.5), (100000, 0.3), (300000, 0.97)):
        idx = np.flatnonzero(rng.random(n) < p)
        if len(idx) == 0:
            idx = np.array([0])
        s, r = AU.rle_encode(idx)
        es, er = OR.rle_encode(idx)
        np.testing.assert_array_equal(s, es); np.testing.assert_array_equal(r, er)
        np.testing.assert_array_equal(AU.rle_decode(s, r), idx)
    # one very long run and duplicate indices (each duplicate starts a new run, like the reference)
    s, r = AU.rle_encode(np.arange(7, 7 + 200000))
    assert s.tolist() == [7] and r.tolist() == [200000]
    dup = np.array([3, 3, 4, 4, 5])
    np.testing.assert_array_equal(AU.rle_encode(dup)[0], OR.rle_encode(dup)[0])
    np.testing.assert_array_equal(AU.rle_encode(dup)[1], OR.rle_encode(dup)[1])
    with pytest.raises(IndexError):
        AU.rle_encode(np.zeros(0, dtype=np.int64))
    with pytest.raises(ValueError):
        AU.rle_decode(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))


def test_box_pairs_golden(hip):
    from empanada_amd import array_utils as AU
    g = load_golden('array_utils')
    for nd in (2, 3):
        a, b = g[f'box{nd}_a'], g[f'box{nd}_b']
        r, c, iou, inter = AU.box_pairs(a, b)
        assert np.all(np.diff(r * len(b) + c) > 0)                      # row-major, unique
        dense_iou = np.zeros((len(a), len(b))); dense_iou[r, c] = iou
        dense_int = np.zeros((len(a), len(b))); dense_int[r, c] = inter
        np.testing.assert_array_equal(dense_iou, g[f'box{nd}_iou'])
        np.testing.assert_array_equal(dense_int, g[f'box{nd}_inter'])
        np.testing.assert_array_equal(AU.box_iou(a, b).toarray(), g[f"box{nd}_iou"])


@pytest.mark.parametrize('k,shape,bias', [(5, (2, 37, 29, 32), False), (3, (1, 8, 6, 8), True),
                                          (5, (3, 64, 64, 72), True), (5, (1, 2, 3, 4), False),
                                          (3, (2, 70, 33, 260), False)])
def test_dwconv_nhwc(hip, k, shape, bias):
    """emp_dwconv_nhwc: bit-exact against the C oracle (same fma chain), and within fp32 rounding of torch's
    conv2d(groups=C), the call the reference makes (blocks.py:27).  Tolerance: |err| <= 1e-5 * (sum |x||w| + 1)."""
    from oracle import dense as OD
    N, H, W, C = shape
    g = torch.Generator().manual_seed(k * 1000 + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.2
    b = torch.randn(C, generator=g) if bias else None
    w_kkc = w.reshape(C, k * k).t().contiguous()
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    got = hip.dwconv_nhwc(xd, w_kkc.cuda(), b.cuda() if bias else None, k)
    assert got.is_contiguous(memory_format=torch.channels_last)
    got_nhwc = got.permute(0, 2, 3, 1).cpu().numpy()
    exp = OD.dwconv_nhwc(x.permute(0, 2, 3, 1).numpy(), w_kkc.numpy(), b.numpy() if bias else None)
    np.testing.assert_array_equal(got_nhwc.view(np.uint32), exp.view(np.uint32))
    ref = torch.nn.functional.conv2d(x, w, b, padding=k // 2, groups=C)
    bound = torch.nn.functional.conv2d(x.abs(), w.abs(), None, padding=k // 2, groups=C) + 1
    assert torch.all((got.cpu() - ref).abs() <= 1e-5 * bound)


def test_dwconv_argument_errors(hip):
    x = torch.zeros(1, 6, 4, 4, device='cuda').contiguous(memory_format=torch.channels_last)
    with pytest.raises(hip.HipError):
        hip.dwconv_nhwc(x, torch.zeros(25, 6, device='cuda'), None, 5)        # C % 4 != 0
    x = torch.zeros(1, 8, 4, 4, device='cuda').contiguous(memory_format=torch.channels_last)
    with pytest.raises(hip.HipError):
        hip.dwconv_nhwc(x, torch.zeros(49, 8, device='cuda'), None, 7)        # unsupported k


@pytest.mark.parametrize('shape,size,layout', [((2, 1, 16, 16), (64, 64), 'nchw'), ((3, 2, 9, 7), (36, 28), 'nhwc'),
                                                ((2, 8, 5, 6), (10, 12), 'nhwc'), ((1, 12, 1, 1), (4, 5), 'nhwc'),
                                                ((2, 3, 7, 5), (7, 5), 'nchw'), ((1, 4, 6, 6), (11, 13), 'nchw')])
def test_upsample_bilinear(hip, shape, size, layout):
    """emp_upsample_bilinear: bit-exact against the numpy oracle; within 1e-6 * max|x| of torch's interpolate."""
    from oracle import dense as OD
    g = torch.Generator().manual_seed(sum(shape) + size[0])
    x = torch.randn(*shape, generator=g)
    xd = x.cuda()
    if layout == 'nhwc':
        xd = xd.contiguous(memory_format=torch.channels_last)
    got = hip.upsample_bilinear(xd, size)
    exp = OD.upsample_bilinear(x.numpy(), size)
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    ref = torch.nn.functional.interpolate(x, size=size, mode='bilinear', align_corners=True)
    assert (got.cpu() - ref).abs().max() <= 1e-6 * x.abs().max()
    # into a channel slice of a wider channels_last buffer (the decoder's concat target)
    N, C = shape[:2]
    buf = torch.full((N, C + 4, size[0], size[1]), -7.0, device='cuda').contiguous(memory_format=torch.channels_last)
    hip.upsample_bilinear(xd, size, out=buf[:, 4:])
    np.testing.assert_array_equal(buf[:, 4:].cpu().numpy().view(np.uint32), exp.view(np.uint32))
    assert torch.all(buf[:, :4] == -7.0)


def test_bn_act_into_channel_slice(hip):
    x = torch.randn(2, 8, 5, 7).cuda().contiguous(memory_format=torch.channels_last)
    sc, sh = torch.rand(8).cuda() + 0.5, torch.randn(8).cuda()
    buf = torch.full((2, 20, 5, 7), -3.0, device='cuda').contiguous(memory_format=torch.channels_last)
    exp = torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    hip.bn_act_nhwc_(x.clone(memory_format=torch.channels_last), sc, sh, None, True, out=buf[:, 8:16])
    assert torch.equal(buf[:, 8:16], exp) and torch.all(buf[:, :8] == -3.0) and torch.all(buf[:, 16:] == -3.0)
    inplace = hip.bn_act_nhwc_(x.clone(memory_format=torch.channels_last), sc, sh, None, True)
    assert torch.equal(inplace, exp)


@pytest.mark.parametrize('cfg', [
    # N, H, W, Cin, Cout, res, relu, out slice, res slice
    (1, 255, 259, 64, 256, True, True, False, False),        # layer1 conv3 + identity + ReLU; last tile is partial
    (2, 192, 192, 64, 256, False, False, False, False),      # layer1 shortcut projection (no residual, no ReLU)
    (1, 256, 258, 128, 256, True, True, True, True),         # K = 128 (two slabs), output / residual = channel slices
    (1, 300, 220, 128, 128, False, True, False, False),      # a single cout group
    (1, 257, 256, 64, 384, True, False, False, False),       # three cout groups (grid of 240 blocks)
])
def test_conv1x1_weight_stationary(hip, cfg):
    """D4b (emp_conv1x1.hip), reached through emp_conv_bn_act_nhwc for short-K pointwise layers with >= 65 536 pixels:
    bit-exact against the C oracle with the K-slab the dispatcher reports (64), and the tiled kernel
    (EMP_CONV_NO_WS is for A/B runs only) is NOT what ran: emp_conv_k_slab_geom says 64, which it never uses."""
    from oracle import dense as OD
    N, H, W, Cin, Cout, use_res, relu, out_slice, res_slice = cfg
    M = N * H * W
    assert hip.conv_k_slab(M, Cout, 1, use_res, Cin, geom=(1, 1, 1, 0), relu=relu) == 64
    assert hip.conv_k_slab(M, Cout, 1, use_res, Cin, geom=(1, 1, 2, 0), relu=relu) in (16, 32)       # strided: tiled kernel
    assert hip.conv_k_slab(1000, Cout, 1, use_res, Cin, geom=(1, 1, 1, 0), relu=relu) in (16, 32)    # too few pixels
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * (1.0 / Cin ** 0.5)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    res = torch.randn(N, Cout, H, W, generator=g) if use_res else None
    w_okkc = w.permute(0, 2, 3, 1).contiguous()
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    resd = None
    if use_res:
        if res_slice:
            wide = torch.zeros(N, Cout + 64, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
            wide[:, 32:32 + Cout] = res.cuda()
            resd = wide[:, 32:32 + Cout]
        else:
            resd = res.cuda().contiguous(memory_format=torch.channels_last)
    out = None
    if out_slice:
        buf = torch.full((N, Cout + 128, H, W), -7.0, device='cuda').contiguous(memory_format=torch.channels_last)
        out = buf[:, 64:64 + Cout]
    got = hip.conv_bn_act_nhwc(xd, w_okkc.cuda(), sc.cuda(), sh.cuda(), resd, relu, 1, 0, 1, out=out)
    exp = OD.conv_bn_act_nhwc(x.permute(0, 2, 3, 1).numpy(), w_okkc.numpy(), sc.numpy(), sh.numpy(),
                              res.permute(0, 2, 3, 1).numpy() if use_res else None, relu, 1, 0, 1, slab=64)
    np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))
    if out_slice:
        assert torch.all(buf[:, :64] == -7.0) and torch.all(buf[:, 64 + Cout:] == -7.0)
    y = torch.nn.functional.conv2d(x, w) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if use_res:
        y = y + res
    if relu:
        y = torch.relu(y)
    bound = torch.nn.functional.conv2d(x.abs(), w.abs()) * sc.view(1, -1, 1, 1)
    assert torch.all((got.cpu() - y).abs() <= 2e-6 * bound + 1e-6)


@pytest.mark.parametrize('cfg', [
    # N, H, W, Cin, Cout, k, stride, pad, dil, res, relu, affine
    (2, 9, 11, 64, 128, 1, 1, 0, 1, True, True, True),
    (1, 12, 10, 32, 40, 3, 1, 1, 1, False, True, True),        # Cout not a tile multiple
    (2, 13, 13, 64, 256, 3, 1, 2, 2, False, True, True),       # dilated (ASPP / layer4)
    (1, 16, 14, 96, 64, 3, 2, 1, 1, False, False, True),       # stride 2
    (3, 7, 5, 32, 32, 1, 2, 0, 1, False, False, False),        # downsample-like, no epilogue
    (1, 10, 10, 64, 130, 3, 1, 6, 6, True, False, True),       # dilation wider than the image border
    (8, 64, 64, 32, 320, 1, 1, 0, 1, False, True, True),       # > 512 blocks: the 16-wide K-slab variant
    (8, 64, 64, 32, 320, 1, 1, 0, 1, True, True, True),        # same with a residual (not prefetchable: Cout % 128)
])
def test_conv_bn_act_nhwc(hip, cfg):
    """emp_conv_bn_act_nhwc: bit-exact against the C oracle (same fma chain); within fp32 rounding of torch's
    conv2d + affine + residual + relu.  Tolerance vs torch: |err| <= 2e-6 * sum|x||w| * |scale| + 1e-6."""
    from oracle import dense as OD
    N, H, W, Cin, Cout, k, stride, pad, dil, use_res, relu, affine = cfg
    g = torch.Generator().manual_seed(Cin * 7 + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k) ** 0.5)
    sc = torch.rand(Cout, generator=g) + 0.5 if affine else None
    sh = torch.randn(Cout, generator=g) if affine else None
    ref = torch.nn.functional.conv2d(x, w, None, stride=stride, padding=pad, dilation=dil)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    w_okkc = w.permute(0, 2, 3, 1).contiguous()
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    resd = res.cuda().contiguous(memory_format=torch.channels_last) if use_res else None
    got = hip.conv_bn_act_nhwc(xd, w_okkc.cuda(), sc.cuda() if affine else None, sh.cuda() if affine else None,
                               resd, relu, stride, pad, dil)
    exp = OD.conv_bn_act_nhwc(x.permute(0, 2, 3, 1).numpy(), w_okkc.numpy(), sc.numpy() if affine else None,
                              sh.numpy() if affine else None,
                              res.permute(0, 2, 3, 1).numpy() if use_res else None, relu, stride, pad, dil,
                              slab=hip.conv_k_slab(ref.shape[0] * ref.shape[2] * ref.shape[3], Cout, 1, use_res))
    np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))
    bound = torch.nn.functional.conv2d(x.abs(), w.abs(), None, stride=stride, padding=pad, dilation=dil)
    y = ref
    if affine:
        y = y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        bound = bound * sc.view(1, -1, 1, 1)
    if use_res:
        y = y + res
    if relu:
        y = torch.relu(y)
    assert torch.all((got.cpu() - y).abs() <= 2e-6 * bound + 1e-6)
    # into a channel slice of a wider buffer
    buf = torch.full((N, Cout + 8, ref.shape[2], ref.shape[3]), 5.0, device='cuda').contiguous(memory_format=torch.channels_last)
    hip.conv_bn_act_nhwc(xd, w_okkc.cuda(), sc.cuda() if affine else None, sh.cuda() if affine else None, resd, relu,
                         stride, pad, dil, out=buf[:, 8:])
    assert torch.equal(buf[:, 8:], got) and torch.all(buf[:, :8] == 5.0)


@pytest.mark.parametrize('cfg', [
    # N, H, W, Cin, Cout, k, stride, pad, dil, residual, relu, affine, k_splits (None = emp_conv_splitk_plan)
    (1, 32, 32, 256, 256, 3, 1, 1, 1, False, True, True, None),     # layer3 conv2 of a 512^2 tile: 32 tiles -> 8 ranges
    (1, 32, 32, 1024, 256, 1, 1, 0, 1, False, True, True, None),    # layer3 conv1
    (1, 32, 32, 256, 1024, 1, 1, 0, 1, True, True, True, 3),        # conv3 + identity
    (1, 16, 24, 64, 48, 3, 2, 1, 1, False, False, False, 2),        # stride 2, no affine, narrow tile, ragged M
    (2, 9, 7, 96, 132, 3, 1, 2, 2, True, True, True, 5),            # dilation, couts past the last tile
    (1, 8, 8, 32, 64, 1, 1, 0, 1, False, False, True, 1),           # one range = the unsplit summation with K-slab 32
])
def test_conv_splitk_bn_act_nhwc(hip, cfg):
    """emp_conv_splitk_bn_act_nhwc (D4c, small launches): bit-exact against the C oracle (partial fmaf chains per K
    range, added in ascending order, separate roundings in the epilogue); deterministic from run to run; within fp32
    rounding of torch's conv2d + affine + residual + relu: |err| <= 2e-6 * sum|x||w| * |scale| + 1e-6."""
    from oracle import dense as OD
    N, H, W, Cin, Cout, k, stride, pad, dil, use_res, relu, affine, ks = cfg
    g = torch.Generator().manual_seed(Cin * 5 + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k) ** 0.5)
    sc = torch.rand(Cout, generator=g) + 0.5 if affine else None
    sh = torch.randn(Cout, generator=g) if affine else None
    ref = torch.nn.functional.conv2d(x, w, None, stride=stride, padding=pad, dilation=dil)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    w_okkc = w.permute(0, 2, 3, 1).contiguous()
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    resd = res.cuda().contiguous(memory_format=torch.channels_last) if use_res else None
    M = ref.shape[0] * ref.shape[2] * ref.shape[3]
    if ks is None:
        ks = hip.conv_splitk_plan(M, Cout, Cin, k, k)
        assert ks >= 2
    args = (xd, w_okkc.cuda(), sc.cuda() if affine else None, sh.cuda() if affine else None, resd, relu, stride, pad, dil)
    got = hip.conv_splitk_bn_act_nhwc(*args, k_splits=ks)
    assert torch.equal(got, hip.conv_splitk_bn_act_nhwc(*args, k_splits=ks))
    exp = OD.conv_splitk_bn_act_nhwc(x.permute(0, 2, 3, 1).numpy(), w_okkc.numpy(), sc.numpy() if affine else None,
                                     sh.numpy() if affine else None,
                                     res.permute(0, 2, 3, 1).numpy() if use_res else None, relu, stride, pad, dil, ks)
    np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))
    if ks == 1:                            # one range: the plain kernel's sum whenever that runs with K-slab 32
        plain = OD.conv_bn_act_nhwc(x.permute(0, 2, 3, 1).numpy(), w_okkc.numpy(), sc.numpy() if affine else None,
                                    sh.numpy() if affine else None, None, relu, stride, pad, dil, slab=32)
        np.testing.assert_array_equal(exp.view(np.uint32), plain.view(np.uint32))
    bound = torch.nn.functional.conv2d(x.abs(), w.abs(), None, stride=stride, padding=pad, dilation=dil)
    y = ref
    if affine:
        y = y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        bound = bound * sc.view(1, -1, 1, 1)
    if use_res:
        y = y + res
    if relu:
        y = torch.relu(y)
    assert torch.all((got.cpu() - y).abs() <= 2e-6 * bound + 1e-6)
    buf = torch.full((N, Cout + 8, ref.shape[2], ref.shape[3]), 5.0, device='cuda').contiguous(memory_format=torch.channels_last)
    hip.conv_splitk_bn_act_nhwc(*args, out=buf[:, 8:], k_splits=ks)
    assert torch.equal(buf[:, 8:], got) and torch.all(buf[:, :8] == 5.0)


@pytest.mark.parametrize('cfg', [
    (2, 9, 11, 2, 72, 1, True),        # RegNetY-6.4GF stage 1: 2 groups of 72, chunks of 24 channels
    (1, 16, 16, 4, 72, 2, True),       # stride 2 (first block of a stage)
    (3, 7, 5, 18, 72, 1, False),       # 18 groups (stage 4 width 1296), no ReLU
    (2, 12, 10, 3, 56, 1, True),       # RegNetY-8GF group width: chunks of 8
    (1, 9, 9, 2, 112, 2, True),        # RegNetY-16GF: chunks of 16, 7 cout tiles
    (1, 20, 20, 5, 8, 1, True),        # narrowest group
    (1, 6, 6, 1, 128, 1, True),        # widest group (8 cout tiles)
    (4, 64, 64, 8, 72, 1, True),       # 1024 blocks: every XCD slot used several times over
])
def test_gconv3x3_bn_act_nhwc(hip, cfg):
    """emp_gconv3x3_bn_act_nhwc (D8): bit-exact against the C oracle (same fma chain over 16x16x4 MFMAs); within
    fp32 rounding of torch's conv2d(groups=G) + affine + relu, |err| <= 2e-6 * sum|x||w| * |scale| + 1e-6; output
    into a channel slice of a wider NHWC buffer leaves the other channels untouched."""
    from oracle import dense as OD
    N, H, W, G, GW, stride, relu = cfg
    C = G * GW
    g = torch.Generator().manual_seed(C * 3 + H + stride)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, GW, 3, 3, generator=g) * (1.0 / (GW * 9) ** 0.5)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    w_okkc = w.permute(0, 2, 3, 1).contiguous()
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    got = hip.gconv3x3_bn_act_nhwc(xd, w_okkc.cuda(), G, sc.cuda(), sh.cuda(), relu, stride)
    exp = OD.gconv3x3_bn_act_nhwc(x.permute(0, 2, 3, 1).numpy(), w_okkc.numpy(), G, sc.numpy(), sh.numpy(), relu, stride)
    assert OD.gconv_chunk(GW) == hip.load().emp_gconv_chunk(GW)
    np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))
    ref = torch.nn.functional.conv2d(x, w, None, stride=stride, padding=1, groups=G)
    bound = torch.nn.functional.conv2d(x.abs(), w.abs(), None, stride=stride, padding=1, groups=G) * sc.view(1, -1, 1, 1)
    y = ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if relu:
        y = torch.relu(y)
    assert torch.all((got.cpu() - y).abs() <= 2e-6 * bound + 1e-6)
    buf = torch.full((N, C + 8, ref.shape[2], ref.shape[3]), 5.0, device='cuda').contiguous(memory_format=torch.channels_last)
    hip.gconv3x3_bn_act_nhwc(xd, w_okkc.cuda(), G, sc.cuda(), sh.cuda(), relu, stride, out=buf[:, 8:])
    assert torch.equal(buf[:, 8:], got) and torch.all(buf[:, :8] == 5.0)
    # no epilogue operands
    got0 = hip.gconv3x3_bn_act_nhwc(xd, w_okkc.cuda(), G, None, None, False, stride)
    exp0 = OD.gconv3x3_bn_act_nhwc(x.permute(0, 2, 3, 1).numpy(), w_okkc.numpy(), G, None, None, False, stride)
    np.testing.assert_array_equal(got0.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp0.view(np.uint32))


@pytest.mark.parametrize('shape', [(2, 1, 64, 64), (1, 1, 37, 45), (3, 5, 32, 48), (2, 3, 17, 9), (1, 16, 8, 8), (0, 1, 4, 4)])
def test_logits_to_prob_kernel(hip, shape):
    """emp_logits_to_prob (D2, engines.py:22-30): against the oracle (same operation order, host expf) and against
    torch's own sigmoid / softmax on the GPU and on the CPU, all within 4 ulp; softmax rows sum to 1 within 3e-7; in
    place (prob aliasing logits) gives the same bits; odd sizes take the scalar path."""
    from oracle import dense as OD
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g) * 4
    got = hip.logits_to_prob(x.cuda())
    assert got.shape == x.shape
    if x.numel() == 0:
        return
    ref_gpu = (torch.sigmoid(x.cuda()) if shape[1] == 1 else torch.softmax(x.cuda(), dim=1)).cpu()
    ref_cpu = torch.sigmoid(x) if shape[1] == 1 else torch.softmax(x, dim=1)
    exp = torch.from_numpy(OD.logits_to_prob(x.numpy()))

    def ulps(a, b):
        return (a.view(torch.int32) - b.view(torch.int32)).abs().max().item()

    g_cpu = got.cpu()
    # sigmoid: a few ulp (tiny probabilities p ~ exp(x) inherit the last-place differences of three exp
    # implementations: device library, host libm, ATen's vectorised Sleef); softmax: the quotient of two rounded sums --
    # compared absolutely (probabilities <= 1)
    if shape[1] == 1:
        u = (ulps(g_cpu, exp), ulps(g_cpu, ref_gpu), ulps(g_cpu, ref_cpu))
        assert max(u) <= 4, u
    else:
        for r in (exp, ref_gpu, ref_cpu):
            assert (g_cpu - r).abs().max().item() <= 3e-7
        assert (g_cpu.sum(dim=1) - 1).abs().max().item() <= 3e-7
        assert int((g_cpu.argmax(dim=1) != ref_cpu.argmax(dim=1)).sum()) == 0
    y = x.cuda().clone()
    hip.logits_to_prob(y, out=y)
    assert torch.equal(y, got)


# ------------------------------------------------------------------------------------------------ D10: PointRend step
@pytest.mark.parametrize('shape', [(2, 1, 16, 20), (1, 3, 9, 7), (3, 1, 128, 128), (1, 5, 33, 1)])
def test_pr_upsample2x(hip, shape):
    """emp_pr_upsample2x: bit-exact against the oracle (same fp32 operations); within rounding of torch's
    F.interpolate(x2, bilinear, align_corners=False) and of calculate_uncertainty (point_rend.py:62-79)."""
    from oracle import dense as OD
    x = torch.randn(shape, generator=torch.Generator().manual_seed(sum(shape))) * 3
    up, unc = hip.pr_upsample2x(x.cuda())
    eu, ec = OD.pr_upsample2x(x.numpy())
    np.testing.assert_array_equal(up.cpu().numpy().view(np.uint32), eu.view(np.uint32))
    np.testing.assert_array_equal(unc.cpu().numpy().view(np.uint32), ec.view(np.uint32))
    ref = torch.nn.functional.interpolate(x.cuda(), scale_factor=2.0, mode='bilinear', align_corners=False)
    assert (up - ref).abs().max().item() <= 2e-6 * x.abs().max().item()


@pytest.mark.parametrize('case', ['random', 'ties', 'all_equal', 'k_is_all', 'k_is_one', 'ragged'])
def test_pr_topk(hip, case):
    """emp_pr_topk: exactly the oracle's selection and order (the k largest, ties at the k-th value to the lowest
    indices), and the same SET as torch.topk wherever the k-th value is unique."""
    from oracle import dense as OD
    g = torch.Generator().manual_seed(len(case))
    N, HW, k = 3, 40000, 8192
    u = -torch.rand((N, HW), generator=g) * 5
    if case == 'ties':
        u = torch.round(u * 4) / 4                       # 21 distinct values: thousands of ties at the threshold
    elif case == 'all_equal':
        u = torch.zeros((N, HW)) - 1.5
    elif case == 'k_is_all':
        N, HW, k = 2, 3000, 3000
        u = u[:N, :HW].contiguous()
    elif case == 'k_is_one':
        k = 1
    elif case == 'ragged':
        N, HW, k = 5, 16384 * 2 + 77, 5000                # three chunks per image, the last almost empty
        u = -torch.rand((N, HW), generator=g) * 5
        u[2] = torch.round(u[2] * 8) / 8
        u[3, 1000:30000] = float('-inf')
    idx = hip.pr_topk(u.cuda(), k).cpu().numpy()
    exp = OD.pr_topk(u.numpy(), k)
    np.testing.assert_array_equal(idx, exp)
    ref = torch.topk(u, k=k, dim=1)
    for n in range(N):
        assert len(set(idx[n].tolist())) == k
        if case in ('random', 'k_is_one'):
            assert set(idx[n].tolist()) == set(ref[1][n].tolist())
        np.testing.assert_array_equal(np.sort(u[n].numpy()[idx[n]]), np.sort(ref[0][n].numpy()))


@pytest.mark.parametrize('cfg', [(2, 12, 10, 256, 1, 300, 2), (1, 7, 9, 64, 3, 63, 4), (3, 16, 16, 128, 2, 1000, 2)])
def test_pr_point_sample_and_scatter(hip, cfg):
    """emp_pr_point_sample: bit-exact against the oracle; within rounding of F.grid_sample(bilinear,
    align_corners=False) -- border points (zero padding) included; X1 carries the coarse channels only.
    emp_pr_scatter: logits[n, c, idx] = points."""
    from oracle import dense as OD
    N, Hf, Wf, CF, C, k, up = cfg
    H, W = Hf * up, Wf * up
    g = torch.Generator().manual_seed(CF + k)
    feat = torch.randn(N, CF, Hf, Wf, generator=g)
    coarse = torch.randn(N, C, Hf, Wf, generator=g)
    idx = torch.stack([torch.randperm(H * W - 2, generator=g)[:k] + 1 for _ in range(N)]).to(torch.int32)   # distinct
    idx[:, 0], idx[:, 1] = 0, H * W - 1                   # corners: three of four neighbours outside the map
    ld = (CF + C + 15) // 16 * 16
    fd = feat.cuda().contiguous(memory_format=torch.channels_last)
    X0, X1 = hip.pr_point_sample(fd, coarse.cuda(), idx.cuda(), H, W, ld)
    exp = OD.pr_point_sample(feat.permute(0, 2, 3, 1).numpy(), coarse.numpy(), idx.numpy(), H, W, ld)
    np.testing.assert_array_equal(X0.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    np.testing.assert_array_equal(X1[:, CF:].cpu().numpy().view(np.uint32), exp[:, CF:].view(np.uint32))
    it = idx.long()
    coords = torch.zeros(N, k, 2)
    coords[:, :, 0] = 0.5 / W + (it % W).float() / float(W)
    coords[:, :, 1] = 0.5 / H + torch.div(it, W, rounding_mode='floor').float() / float(H)
    ref = torch.nn.functional.grid_sample(feat, 2.0 * coords.unsqueeze(2) - 1.0, mode='bilinear', align_corners=False)
    ref = ref.squeeze(3).permute(0, 2, 1).reshape(N * k, CF)
    assert (X0[:, :CF].cpu() - ref).abs().max().item() <= 4e-6 * feat.abs().max().item()
    logits = torch.zeros(N, C, H, W, device='cuda')
    pts = torch.randn(N * k, C, generator=g).cuda()
    hip.pr_scatter(pts, idx.cuda(), logits)
    want = torch.zeros(N, C, H * W).scatter_(2, it.unsqueeze(1).expand(-1, C, -1), pts.cpu().view(N, k, C).permute(0, 2, 1))
    assert torch.equal(logits.cpu().view(N, C, -1), want)


@pytest.mark.parametrize('C', [1, 3])
def test_pointrend_head_hip_vs_library(hip, C):
    """PointRendSemSegHead on the D10 kernels against the same module on the library calls the reference makes
    (interpolate / topk / grid_sample / Conv1d / scatter_, point_rend.py:241-269), both on the GPU: the refined logits
    agree to fp32 rounding except where a rounding difference swaps points at the k-th uncertainty; the point MLP is
    bit-exact against the oracle's convolution (summation order of D4)."""
    from oracle import dense as OD
    from empanada_amd.models.panoptic_deeplab import PointRendSemSegHead
    from empanada_amd.models import synthesize_weights
    torch.manual_seed(C)
    head = synthesize_weights(PointRendSemSegHead(256, C, subdivision_num_points=2048)).eval().cuda()
    N, hf, wf = 2, 24, 20
    feats = torch.randn(N, 256, hf, wf, device='cuda').contiguous(memory_format=torch.channels_last)
    coarse = torch.randn(N, C, hf, wf, device='cuda')
    with torch.no_grad():
        head.hip_ops = False
        ref = head(coarse, feats)['sem_seg_logits']
        head.hip_ops = True
        got = head(coarse, feats)['sem_seg_logits']
        again = head(coarse, feats)['sem_seg_logits']
    assert got.shape == ref.shape == (N, C, 4 * hf, 4 * wf) and torch.equal(got, again)
    bad = (got - ref).abs() > 1e-4 * ref.abs().max() + 1e-5
    assert bad.float().mean().item() < 2e-3, bad.float().mean().item()
    # first MLP layer against the oracle's convolution on the sampled matrix
    w = head._hip_weights()
    up, unc = hip.pr_upsample2x(coarse)
    idx = hip.pr_topk(unc, min(4 * hf * wf, 2048))
    X0, X1 = hip.pr_point_sample(feats, coarse, idx, 2 * hf, 2 * wf, head._hip_ld)
    hip.conv_bn_act_nhwc(hip.as_pixels(X0), w[0], None, w[1], None, True, out=hip.as_pixels(X1, 256))
    P = X0.shape[0]
    exp = OD.conv_bn_act_nhwc(X0.cpu().numpy().reshape(1, P, 1, -1), w[0].cpu().numpy(), None, w[1].cpu().numpy(), None,
                              True, slab=hip.conv_k_slab(P, 256, 1, False, head._hip_ld))
    np.testing.assert_array_equal(X1[:, :256].cpu().numpy().view(np.uint32), exp.reshape(P, 256).view(np.uint32))


@pytest.mark.parametrize('shape', [(2, 64, 64), (1, 37, 45), (3, 130, 70), (1, 7, 9), (2, 256, 320), (1, 1, 1)])
def test_stem_conv7_bn_relu_maxpool(hip, shape):
    """emp_stem_conv7_bn_relu_maxpool (D9): bit-exact against the oracle (fma chain over the 49 taps in raster order,
    then the D7 epilogue); within fp32 rounding of torch's conv2d -> affine -> relu -> max_pool2d:
    |err| <= 2e-6 * sum|x||w| * |scale| + 1e-6.  Odd sizes: partial tiles, windows cut by every border."""
    from oracle import dense as OD
    N, H, W = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(N, 1, H, W, generator=g)
    w = torch.randn(64, 1, 7, 7, generator=g) * (1.0 / 7)
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.5
    w_tc = w[:, 0].reshape(64, 49).t().contiguous()
    got = hip.stem_conv7_bn_relu_maxpool(x.cuda(), w_tc.cuda(), sc.cuda(), sh.cuda())
    exp = OD.stem_conv7_bn_relu_maxpool(x[:, 0].numpy(), w_tc.numpy(), sc.numpy(), sh.numpy())
    np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))
    conv = torch.nn.functional.conv2d(x, w, stride=2, padding=3)
    ref = torch.nn.functional.max_pool2d(torch.relu(conv * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1)
    bound = torch.nn.functional.max_pool2d(torch.nn.functional.conv2d(x.abs(), w.abs(), stride=2, padding=3)
                                           * sc.view(1, -1, 1, 1), 3, 2, 1)
    assert got.shape == ref.shape and torch.all((got.cpu() - ref).abs() <= 2e-6 * bound + 1e-6)
    # the two-kernel path it replaces (MIOpen convolution + emp_bn_relu_maxpool_nhwc) agrees to the same bound
    y = conv.cuda().contiguous(memory_format=torch.channels_last)
    two = hip.bn_relu_maxpool_nhwc(y, sc.cuda(), sh.cuda())
    assert torch.all((got - two).abs().cpu() <= 2e-6 * bound + 1e-6)


@pytest.mark.parametrize('gw', list(range(8, 129, 8)))
def test_gconv_every_group_width(hip, gw):
    """every (chunk, cout-tile) instantiation of gconv3x3_f32_kernel that ships -- group widths 8, 16, .., 128 -- bit-exact
    against the oracle, stride 1 and 2, on a shape with partial pixel tiles"""
    from oracle import dense as OD
    G, N, H, W = 2, 1, 11, 13
    C = G * gw
    g = torch.Generator().manual_seed(gw)
    x = torch.randn(N, C, H, W, generator=g)
    w = (torch.randn(C, gw, 3, 3, generator=g) * (1.0 / (gw * 9) ** 0.5)).permute(0, 2, 3, 1).contiguous()
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    for stride in (1, 2):
        got = hip.gconv3x3_bn_act_nhwc(xd, w.cuda(), G, sc.cuda(), sh.cuda(), True, stride)
        exp = OD.gconv3x3_bn_act_nhwc(x.permute(0, 2, 3, 1).numpy(), w.numpy(), G, sc.numpy(), sh.numpy(), True, stride)
        np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))


def test_gconv_and_gate_argument_errors(hip):
    """emp_gconv3x3_bn_act_nhwc / the gate epilogue refuse what they cannot compute (EMP_EINVAL -> HipError, nothing
    launched); an empty batch is a no-op."""
    x = torch.zeros(1, 36, 8, 8, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.zeros(36, 3, 3, 12, device='cuda')
    with pytest.raises(hip.HipError):
        hip.gconv3x3_bn_act_nhwc(x, w, 3)                                    # group width 12 is not a multiple of 8
    x = torch.zeros(1, 32, 8, 8, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.zeros(32, 3, 3, 16, device='cuda')
    with pytest.raises(hip.HipError):
        hip.gconv3x3_bn_act_nhwc(x, w, 2, stride=3)                          # stride 1 or 2 only
    x0 = torch.zeros(0, 32, 8, 8, device='cuda').contiguous(memory_format=torch.channels_last)
    assert hip.gconv3x3_bn_act_nhwc(x0, w, 2).shape == (0, 32, 8, 8)
    w1 = torch.zeros(16, 1, 1, 32, device='cuda')
    with pytest.raises(hip.HipError):
        hip.conv_bn_act_nhwc(x, w1, None, None, None, 'gate')                # the gate needs the gated tensor
    x24 = torch.zeros(1, 24, 8, 8, device='cuda').contiguous(memory_format=torch.channels_last)
    with pytest.raises(hip.HipError):
        hip.conv_bn_act_nhwc(x24, torch.zeros(16, 1, 1, 24, device='cuda'))   # Cin % 16 != 0


@pytest.mark.parametrize('cfg', [
    (2, 16, 16, 144, 144, False, True),      # RegNetY widths: Cin a multiple of 16 only -> 16-wide K-slabs
    (2, 16, 16, 144, 144, True, True),       # with the shortcut (no residual prefetch on this path)
    (1, 8, 8, 1296, 1296, True, True),
    (8, 64, 64, 144, 48, False, True),       # squeeze conv, Cout padded 36 -> 48 by the host
    (1, 12, 12, 48, 144, False, False),
])
def test_conv_cin_multiple_of_16(hip, cfg):
    """emp_conv_bn_act_nhwc with Cin % 32 != 0: bit-exact against the oracle at S = emp_conv_k_slab_cin(..) == 16."""
    from oracle import dense as OD
    N, H, W, Cin, Cout, use_res, relu = cfg
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * (1.0 / Cin ** 0.5)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    res = torch.randn(N, Cout, H, W, generator=g) if use_res else None
    w_okkc = w.permute(0, 2, 3, 1).contiguous()
    slab = hip.conv_k_slab(N * H * W, Cout, 1, use_res, Cin)
    assert slab == 16
    got = hip.conv_bn_act_nhwc(x.cuda().contiguous(memory_format=torch.channels_last), w_okkc.cuda(), sc.cuda(), sh.cuda(),
                               res.cuda().contiguous(memory_format=torch.channels_last) if use_res else None, relu)
    exp = OD.conv_bn_act_nhwc(x.permute(0, 2, 3, 1).numpy(), w_okkc.numpy(), sc.numpy(), sh.numpy(),
                              res.permute(0, 2, 3, 1).numpy() if use_res else None, relu, slab=slab)
    np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize('cfg', [(2, 16, 16, 48, 144), (1, 8, 8, 80, 288), (8, 64, 64, 144, 576), (1, 4, 4, 336, 1296),
                                 (2, 16, 16, 64, 256), (8, 64, 64, 96, 128)])
def test_conv_gate_epilogue(hip, cfg):
    """relu == 2 (squeeze-excite gate, blocks.py:35-50): out = x * sigmoid(conv(s) + b).  The accumulator is
    bit-exact (same kernel); expf is the device library's, so the gate is compared with the oracle (glibc expf) and
    with torch's CPU sigmoid within a few ulp of the result: |err| <= 5e-7 * |x|."""
    from oracle import dense as OD
    N, H, W, Cin, Cout = cfg
    g = torch.Generator().manual_seed(Cin + Cout)
    s_in = torch.relu(torch.randn(N, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    x = torch.randn(N, Cout, H, W, generator=g) * 3
    w_okkc = w.permute(0, 2, 3, 1).contiguous()
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    got = hip.conv_bn_act_nhwc(s_in.cuda().contiguous(memory_format=torch.channels_last), w_okkc.cuda(), None, b.cuda(),
                               xd, 'gate').cpu()
    slab = hip.conv_k_slab(N * H * W, Cout, 1, False, Cin)      # the gate variant never takes the residual-prefetch plan
    exp = torch.from_numpy(OD.conv_bn_act_nhwc(s_in.permute(0, 2, 3, 1).numpy(), w_okkc.numpy(), None, b.numpy(),
                                               x.permute(0, 2, 3, 1).numpy(), 'gate', slab=slab)).permute(0, 3, 1, 2)
    assert torch.all((got - exp).abs() <= 5e-7 * x.abs())
    ref = x * torch.sigmoid(torch.nn.functional.conv2d(s_in, w, b))
    assert torch.all((got - ref).abs() <= 2e-5 * x.abs())      # conv rounding (K up to 336) inside the sigmoid's slope <= 1/4


@pytest.mark.parametrize('cfg', [(2, 9, 11, 64, 128, 1), (1, 12, 10, 32, 40, 2), (2, 13, 8, 64, 132, 6),
                                 (1, 5, 7, 96, 64, 4), (3, 16, 16, 32, 256, 3)])
def test_winograd_conv(hip, cfg):
    """Winograd F(2x2,3x3) path (D5): bit-exact against the oracle restatement; every output pixel written exactly
    once; within 1e-5 * sum|x||w| (+1e-6) of torch's conv2d + affine + relu."""
    from oracle import dense as OD
    N, H, W, Cin, Cout, dil = cfg
    g = torch.Generator().manual_seed(Cin + Cout + dil + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / (Cin * 9) ** 0.5)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    tiles = hip.wino_tiles(N, H, W, dil)
    # the tiles partition the output pixels
    seen = np.zeros((N, H, W), dtype=np.int32)
    for a in range(2):
        for b in range(2):
            yy, xx = tiles[:, 1] + dil + a * dil, tiles[:, 2] + dil + b * dil
            ok = (yy < H) & (xx < W)
            np.add.at(seen, (tiles[ok, 0], yy[ok], xx[ok]), 1)
    assert np.all(seen == 1)
    U = hip.wino_filter_transform(w)
    np.testing.assert_array_equal(U.numpy().view(np.uint32), OD.wino_filter_transform(w.numpy()).view(np.uint32))
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    got = hip.wino_conv_bn_act(xd, U.cuda(), torch.from_numpy(tiles).cuda(), dil, sc.cuda(), sh.cuda(), True)
    unfused = hip.wino_conv_bn_act(xd, U.cuda(), torch.from_numpy(tiles).cuda(), dil, sc.cuda(), sh.cuda(), True,
                                   fused=False)
    xn, wn = x.permute(0, 2, 3, 1).numpy(), w.numpy()
    # loader-fused input transform (always 32-channel slabs) and separate transform + batched GEMM (slab by K)
    exp = OD.wino_conv_bn_act(xn, wn, tiles, dil, sc.numpy(), sh.numpy(), True, slab=32)
    np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))
    slab = hip.conv_k_slab(len(tiles), Cout, 16)
    exp_u = exp if slab == 32 else OD.wino_conv_bn_act(xn, wn, tiles, dil, sc.numpy(), sh.numpy(), True, slab=slab)
    np.testing.assert_array_equal(unfused.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp_u.view(np.uint32))
    ref = torch.relu(torch.nn.functional.conv2d(x, w, None, padding=dil, dilation=dil) * sc.view(1, -1, 1, 1)
                     + sh.view(1, -1, 1, 1))
    bound = torch.nn.functional.conv2d(x.abs(), w.abs(), None, padding=dil, dilation=dil) * sc.view(1, -1, 1, 1)
    assert torch.all((got.cpu() - ref).abs() <= 1e-5 * bound + 1e-6)


@pytest.mark.parametrize('shape', [(6, 20, 37), (5, 16, 32), (3, 33, 7)])
def test_device_volume_feeder(hip, shape):
    """emp_slices_to_input: all three planes of a resident uint8 volume, bit-exact against the numpy statement of
    albumentations' Normalize formula (fp32 (x - mean*255) * (1/(std*255))) + factor_pad."""
    from empanada_amd.data import DeviceVolume, normalize_constants
    rng = np.random.default_rng(shape[0])
    vol = rng.integers(0, 256, shape, dtype=np.uint8)
    mean, std = 0.508979, 0.148561
    dv = DeviceVolume(vol, mean, std, factor=16)
    m255, inv = normalize_constants(mean, std)
    for axis, ax in (('xy', 0), ('xz', 1), ('yz', 2)):
        planes = np.moveaxis(vol, ax, 0)
        n, h, w = planes.shape
        hp, wp = dv.padded_shape(axis)
        assert hp % 16 == 0 and wp % 16 == 0 and hp >= h and wp >= w and dv.n_slices(axis) == n
        exp = np.zeros((n, 1, hp, wp), dtype=np.float32)
        exp[:, 0, :h, :w] = (planes.astype(np.float32) - np.float32(m255)) * np.float32(inv)
        got = torch.cat([b for _, b in dv.batches(axis, 4)], dim=0).cpu().numpy()
        np.testing.assert_array_equal(got.view(np.uint32), exp.view(np.uint32))
        lo, hi = 1, n - 1
        np.testing.assert_array_equal(dv.batch(axis, lo, hi).cpu().numpy(), exp[lo:hi])


@pytest.mark.parametrize('cfg', [(2, 7, 9, 256, 1), (1, 5, 6, 256, 2), (3, 4, 4, 64, 4), (1, 9, 8, 320, 3),
                                 (2, 16, 17, 128, 2)])
def test_pointwise_out(hip, cfg):
    """emp_pointwise_out_nhwc: bit-exact against the numpy restatement of its summation order; within fp32 rounding
    of torch's conv2d (1e-5 * sum|x||w| + 1e-6)."""
    from oracle import dense as OD
    N, H, W, C, Cout = cfg
    g = torch.Generator().manual_seed(C + Cout + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Cout, C, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    got = hip.pointwise_out_nhwc(xd, w.cuda(), b.cuda())
    assert got.is_contiguous() and got.shape == (N, Cout, H, W)
    exp = OD.pointwise_out_nhwc(x.permute(0, 2, 3, 1).numpy(), w.numpy(), b.numpy())
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    ref = torch.nn.functional.conv2d(x, w.view(Cout, C, 1, 1), b)
    bound = torch.nn.functional.conv2d(x.abs(), w.abs().view(Cout, C, 1, 1))
    assert torch.all((got.cpu() - ref).abs() <= 1e-5 * bound + 1e-6)
    nob = hip.pointwise_out_nhwc(xd, w.cuda(), None)
    np.testing.assert_array_equal(nob.cpu().numpy().view(np.uint32),
                                  OD.pointwise_out_nhwc(x.permute(0, 2, 3, 1).numpy(), w.numpy(), None).view(np.uint32))


@pytest.mark.parametrize('shape', [(2, 64, 16, 16), (1, 8, 7, 9), (3, 4, 1, 5), (1, 64, 33, 32)])
def test_bn_relu_maxpool(hip, shape):
    from oracle import dense as OD
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(N, C, H, W, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    got = hip.bn_relu_maxpool_nhwc(x.cuda().contiguous(memory_format=torch.channels_last), sc.cuda(), sh.cuda())
    exp = OD.bn_relu_maxpool_nhwc(x.permute(0, 2, 3, 1).numpy(), sc.numpy(), sh.numpy())
    np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))
    ref = torch.nn.functional.max_pool2d(torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize('cfg', [(2, 9, 11, 64, 128, 1), (1, 12, 10, 32, 40, 2), (2, 13, 8, 64, 132, 6),
                                 (1, 5, 7, 96, 64, 4), (2, 16, 16, 32, 256, 2), (8, 64, 64, 32, 64, 1)])
def test_winograd4_conv(hip, cfg):
    """Winograd F(4x4,3x3) path (D5b): bit-exact against the oracle restatement; tiles partition the outputs; within
    the stated 2e-5 * sum|x||w| (+1e-6) of torch's conv2d + affine + relu."""
    from oracle import dense as OD
    N, H, W, Cin, Cout, dil = cfg
    g = torch.Generator().manual_seed(Cin + Cout + dil + H + 4)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / (Cin * 9) ** 0.5)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    tiles = hip.wino_tiles(N, H, W, dil, m=4)
    seen = np.zeros((N, H, W), dtype=np.int32)
    for a in range(4):
        for b in range(4):
            yy, xx = tiles[:, 1] + dil + a * dil, tiles[:, 2] + dil + b * dil
            ok = (yy < H) & (xx < W)
            np.add.at(seen, (tiles[ok, 0], yy[ok], xx[ok]), 1)
    assert np.all(seen == 1)
    U = hip.wino4_filter_transform(w)
    np.testing.assert_array_equal(U.numpy().view(np.uint32), OD.wino4_filter_transform(w.numpy()).view(np.uint32))
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    got = hip.wino4_conv_bn_act(xd, U.cuda(), torch.from_numpy(tiles).cuda(), dil, sc.cuda(), sh.cuda(), True)
    exp = OD.wino4_conv_bn_act(x.permute(0, 2, 3, 1).numpy(), w.numpy(), tiles, dil, sc.numpy(), sh.numpy(), True,
                               slab=hip.conv_k_slab(len(tiles), Cout, 36))
    np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))
    ref = torch.relu(torch.nn.functional.conv2d(x, w, None, padding=dil, dilation=dil) * sc.view(1, -1, 1, 1)
                     + sh.view(1, -1, 1, 1))
    bound = torch.nn.functional.conv2d(x.abs(), w.abs(), None, padding=dil, dilation=dil) * sc.view(1, -1, 1, 1)
    assert torch.all((got.cpu() - ref).abs() <= 2e-5 * bound + 1e-6)


@pytest.mark.parametrize('cfg', [(2, 9, 11, 64, 128, 1), (1, 12, 10, 32, 40, 2), (2, 32, 32, 32, 64, 6),
                                 (1, 5, 7, 96, 64, 4)])
def test_winograd3_conv(hip, cfg):
    """Winograd F(3x3,3x3) path (D5c): bit-exact against the oracle restatement; tiles partition the outputs; within
    2e-5 * sum|x||w| (+1e-6) of torch's conv2d + affine + relu."""
    from oracle import dense as OD
    N, H, W, Cin, Cout, dil = cfg
    g = torch.Generator().manual_seed(Cin + Cout + dil + H + 3)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (1.0 / (Cin * 9) ** 0.5)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    tiles = hip.wino_tiles(N, H, W, dil, m=3)
    seen = np.zeros((N, H, W), dtype=np.int32)
    for a in range(3):
        for b in range(3):
            yy, xx = tiles[:, 1] + dil + a * dil, tiles[:, 2] + dil + b * dil
            ok = (yy < H) & (xx < W)
            np.add.at(seen, (tiles[ok, 0], yy[ok], xx[ok]), 1)
    assert np.all(seen == 1)
    U = hip.wino3_filter_transform(w)
    np.testing.assert_array_equal(U.numpy().view(np.uint32), OD.wino3_filter_transform(w.numpy()).view(np.uint32))
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    got = hip.wino3_conv_bn_act(xd, U.cuda(), torch.from_numpy(tiles).cuda(), dil, sc.cuda(), sh.cuda(), True)
    exp = OD.wino3_conv_bn_act(x.permute(0, 2, 3, 1).numpy(), w.numpy(), tiles, dil, sc.numpy(), sh.numpy(), True,
                               slab=hip.conv_k_slab(len(tiles), Cout, 25))
    np.testing.assert_array_equal(got.permute(0, 2, 3, 1).cpu().numpy().view(np.uint32), exp.view(np.uint32))
    ref = torch.relu(torch.nn.functional.conv2d(x, w, None, padding=dil, dilation=dil) * sc.view(1, -1, 1, 1)
                     + sh.view(1, -1, 1, 1))
    bound = torch.nn.functional.conv2d(x.abs(), w.abs(), None, padding=dil, dilation=dil) * sc.view(1, -1, 1, 1)
    assert torch.all((got.cpu() - ref).abs() <= 2e-5 * bound + 1e-6)


@pytest.mark.parametrize('cfg', [(2, 9, 11, 64, 256, 1, 2), (1, 7, 5, 32, 128, 1, 1), (2, 8, 8, 64, 256, 3, 4),
                                 (8, 64, 64, 32, 256, 1, 2)])
def test_conv_bn_act_with_projection(hip, cfg):
    """emp_conv_bn_act_proj_nhwc (head pointwise conv + BN + ReLU + last 1x1 conv in one launch): bit-exact against
    the oracle restatement, deterministic across runs (atomics over at most two tiles), activation optional; within
    1e-5 * (sum|x||w| * |scale| * sum|proj_w| ...) of the torch ops."""
    from oracle import dense as OD
    N, H, W, Cin, Cout, k, n = cfg
    g = torch.Generator().manual_seed(Cin + Cout + k + n)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (1.0 / (Cin * k * k) ** 0.5)
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    pw, pb = torch.randn(n, Cout, generator=g) * 0.1, torch.randn(n, generator=g)
    w_okkc = w.permute(0, 2, 3, 1).contiguous()
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    args = (xd, w_okkc.cuda(), sc.cuda(), sh.cuda(), True, pw.cuda(), pb.cuda(), 1, k // 2, 1)
    got = hip.conv_bn_act_proj_nhwc(*args)
    got2, act = hip.conv_bn_act_proj_nhwc(*args, keep=True)
    assert torch.equal(got, got2) and torch.equal(got, hip.conv_bn_act_proj_nhwc(*args))
    plain = hip.conv_bn_act_nhwc(xd, w_okkc.cuda(), sc.cuda(), sh.cuda(), None, True, 1, k // 2, 1)
    assert torch.equal(act, plain)
    M = N * H * W
    exp = OD.conv_bn_act_proj_nhwc(x.permute(0, 2, 3, 1).numpy(), w_okkc.numpy(), sc.numpy(), sh.numpy(), True,
                                   pw.numpy(), pb.numpy(), 1, k // 2, 1, slab=hip.conv_k_slab(M, Cout))
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), exp.view(np.uint32))
    y = torch.relu(torch.nn.functional.conv2d(x, w, None, padding=k // 2) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref = torch.nn.functional.conv2d(y, pw.view(n, Cout, 1, 1), pb)
    bound = torch.nn.functional.conv2d(y.abs() + 1, pw.abs().view(n, Cout, 1, 1))
    assert torch.all((got.cpu() - ref).abs() <= 1e-5 * bound + 1e-6)


# ------------------------------------------------------------------------------------------------ emp_tracks.hip
def _random_run_table(rng, D, H, W, n_inst):
    """a run table like emp_runs_extract / emp_runs_label produce (raster order, runs of a component share a slice),
    with adjacent and row-wrapping runs so that the merge rule has work, plus components without an instance"""
    r_start, r_len, r_comp, c_slice = [], [], [], []
    for d in range(D):
        for y in range(H):
            x = 0
            while x < W:
                if rng.random() < 0.35:
                    x += int(rng.integers(1, 6))
                    continue
                ln = int(min(W - x, rng.integers(1, 9)))
                r_start.append(y * W + x)
                r_len.append(ln)
                r_comp.append(len(c_slice))
                c_slice.append(d)
                x += ln                                   # the next run may start right here: contiguous runs
    n_comp = len(c_slice)
    comp_inst = rng.integers(-1, n_inst, n_comp).astype(np.int32)
    return (np.array(r_start, np.int32), np.array(r_len, np.int32), np.array(r_comp, np.int32),
            np.array(c_slice, np.int32), comp_inst)


@pytest.mark.gpu
@pytest.mark.parametrize('axis', ['xy', 'xz'])
def test_track_lift_sort_offsets_expand_clip(axis):
    from empanada_amd import _hip
    from empanada_amd.inference import device_tracks as DT
    from oracle import tracks as OT
    rng = np.random.default_rng(5)
    for D, H, W, n_inst in ((3, 4, 64, 3), (7, 9, 33, 12), (1, 1, 200, 2), (5, 6, 16, 4000)):
        r_start, r_len, r_comp, c_slice, comp_inst = _random_run_table(rng, D, H, W, n_inst)
        shape3d = (D + 2, H, W) if axis == 'xy' else (H, D + 2, W)
        Z, Y, X = shape3d
        t = _hip.RunTable()
        t.D, t.H, t.W, t.n_runs, t.n_comp = D, H, W, len(r_start), len(c_slice)
        dev = lambda a: torch.from_numpy(a).cuda()
        t.r_start, t.r_len, t.r_comp, t.c_slice = dev(r_start), dev(r_len), dev(r_comp), dev(c_slice)
        base = 7
        key, st, ln, n = DT.lift_runs(t, comp_inst, axis, shape3d, slice0=2, inst_base=base)
        ek, el = OT.lift_xy_xz(0 if axis == 'xy' else 1, r_start, r_len, r_comp, c_slice, comp_inst, H, W, Y, X, 2, base)
        ek, el = OT.sort_runs(ek, el)
        assert n == len(ek) and n > 0
        np.testing.assert_array_equal(key[:n].cpu().numpy().view(np.uint64), ek)
        np.testing.assert_array_equal(ln[:n].cpu().numpy(), el)
        np.testing.assert_array_equal(st[:n].cpu().numpy(), (ek & np.uint64((1 << 40) - 1)).astype(np.int64))
        # offsets / expand
        top = base + n_inst
        off = torch.empty((top + 1,), dtype=torch.int64, device='cuda')
        _hip.call('emp_track_offsets', key.data_ptr(), n, top, off.data_ptr(), _hip.stream())
        eo = OT.offsets(ek, top)
        np.testing.assert_array_equal(off.cpu().numpy(), eo)
        val = rng.integers(0, 100, top).astype(np.int32)
        out = torch.empty((n,), dtype=torch.int32, device='cuda')
        _hip.call('emp_track_expand', off.data_ptr(), dev(val).data_ptr(), top, n, out.data_ptr(), _hip.stream())
        np.testing.assert_array_equal(out.cpu().numpy(), OT.expand(eo, val, n))
        # clip to a slab
        lo, hi = int(0.3 * Z * Y * X), int(0.7 * Z * Y * X)
        ck, cl, cn = DT.clip_runs(key, ln, n, lo, hi)
        xk, xl = OT.clip(ek, el, lo, hi)
        assert cn == len(xk)
        np.testing.assert_array_equal(ck[:cn].cpu().numpy().view(np.uint64), xk)
        np.testing.assert_array_equal(cl[:cn].cpu().numpy(), xl)


@pytest.mark.gpu
def test_track_lift_yz_and_touch_merge():
    from empanada_amd import _hip
    from empanada_amd.inference import device_tracks as DT
    from oracle import tracks as OT
    rng = np.random.default_rng(9)
    for Z, Y, Xl, X, x0 in ((3, 4, 8, 8, 0), (5, 6, 7, 20, 9), (2, 3, 16, 16, 0)):
        # a yz stack: Xl slices of (Z, Y) pixels; instances drawn as boxes so that full rows occur (runs that touch
        # across row ends must merge)
        n_inst = 6
        vol = np.zeros((Z, Y, Xl), dtype=np.int64)
        for i in range(n_inst):
            z0, y0, a0 = rng.integers(0, Z), rng.integers(0, Y), rng.integers(0, Xl)
            vol[z0:z0 + rng.integers(1, 3), y0:y0 + rng.integers(1, 4), a0:] = i + 1
        vol[0, :, :] = 1                                                   # whole rows: touch across row ends
        stack = np.ascontiguousarray(np.moveaxis(vol, 2, 0))               # (Xl, Z, Y) slices
        pan = torch.from_numpy((stack + (stack > 0) * 1000).astype(np.int32)).cuda().view(torch.uint32)
        table = _hip.extract_runs(pan, 1000, [])                           # plain classes: one component per value
        val = table.r_val.cpu().numpy()[table.c_first.cpu().numpy()].astype(np.int64)
        comp_inst = (val - 1001).astype(np.int32)
        key, st, ln, n = DT.lift_runs(table, comp_inst, 'yz', (Z, Y, X), slice0=x0, inst_base=3)
        ek, el = OT.lift_yz(vol, X, x0, 3)
        ek, el = OT.sort_runs(ek, el, merge_touching=True)
        assert n == len(ek)
        np.testing.assert_array_equal(key[:n].cpu().numpy().view(np.uint64), ek)
        np.testing.assert_array_equal(ln[:n].cpu().numpy(), el)
        assert int(ln[:n].sum().item()) == int((vol > 0).sum())


@pytest.mark.gpu
def test_triplets_reduce():
    from empanada_amd import _hip
    from oracle import tracks as OT
    rng = np.random.default_rng(3)
    for n in (1, 17, 5000, 200000):
        trip = np.stack([rng.integers(0, 50, n), rng.integers(0, 60, n), rng.integers(1, 300, n)], axis=1).astype(np.int32)
        got = _hip.reduce_triplets(torch.from_numpy(trip).cuda()).cpu().numpy().astype(np.int64)
        np.testing.assert_array_equal(got, OT.reduce_triplets(trip))
    empty = torch.zeros((0, 3), dtype=torch.int32, device='cuda')
    assert _hip.reduce_triplets(empty).shape[0] == 0
