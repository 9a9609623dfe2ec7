"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

TEST INFRASTRUCTURE ONLY.  Run as `python -m oracle.gen_golden` from the repo root when
/root/reference is present; the GPU box never runs this (it only reads the committed
fixtures).  The reference is imported from where it lies (read-only, no bytecode
written); nothing of it is copied.

Two absent third-party packages are stood in for in-process, before the import
(SURVEY.md Appendix A):
  * numba            -> `jit`/`njit` are the identity decorator (numba only compiles the
                        plain-Python bodies; semantics are the Python semantics).
  * skimage.measure  -> `label` (multi-value, full connectivity, raster-order ids, built on
                        scipy.ndimage.label -- an implementation independent of
                        oracle/c/oracle_kernels.c) and `regionprops` (.label/.bbox/.coords).
  * zarr             -> empty module with an `Array` type (patterns.py:210 isinstance only).
Fixtures hold DATA only: seeded inputs and the reference's outputs.
"""
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


# ----------------------------------------------------------------------------- stand-ins
def _install_standins():
    sys.dont_write_bytecode = True
    nb = types.ModuleType('numba')

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb.jit = nb.njit = jit
    sys.modules['numba'] = nb

    from scipy import ndimage as ndi

    def label(seg, *a, **k):
        seg = np.asarray(seg)
        out = np.zeros(seg.shape, dtype=np.int64)
        st = np.ones((3,) * seg.ndim, dtype=bool)
        firsts = []
        for v in np.unique(seg):
            if v == 0:
                continue
            lab, n = ndi.label(seg == v, structure=st)
            flat = lab.ravel()
            for j in range(1, n + 1):
                firsts.append((int(np.flatnonzero(flat == j)[0]), v, j))
        firsts.sort()
        cache = {}
        for new, (_, v, j) in enumerate(firsts, 1):
            if v not in cache:
                cache[v] = ndi.label(seg == v, structure=st)[0]
            out[cache[v] == j] = new
        return out

    class _RP:
        def __init__(self, lab, sl, coords):
            self.label = int(lab)
            self.bbox = tuple(int(s.start) for s in sl) + tuple(int(s.stop) for s in sl)
            self.coords = coords

    def regionprops(label_image, *a, **k):
        label_image = np.asarray(label_image)
        out = []
        objs = ndi.find_objects(label_image.astype(np.int64))
        for lab, sl in enumerate(objs, 1):
            if sl is None:
                continue
            sub = label_image[sl] == lab
            coords = np.stack(np.nonzero(sub), axis=1) + np.array([s.start for s in sl])
            out.append(_RP(lab, sl, coords))
        return out

    sk = types.ModuleType('skimage')
    me = types.ModuleType('skimage.measure')
    me.label = label
    me.regionprops = regionprops
    sk.measure = me
    sys.modules['skimage'] = sk
    sys.modules['skimage.measure'] = me

    z = types.ModuleType('zarr')
    z.Array = type('Array', (), {})
    sys.modules['zarr'] = z
    sys.path.insert(0, REF)


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def _pack_rle_seg(rle_seg):
    """rle_seg {class: {label: attrs}} -> flat arrays (dict order preserved)."""
    cls, lab, box, off, starts, runs = [], [], [], [0], [], []
    for c, insts in rle_seg.items():
        for l, a in insts.items():
            cls.append(c)
            lab.append(l)
            box.append(a['box'])
            starts.append(np.asarray(a['starts'], dtype=np.int64))
            runs.append(np.asarray(a['runs'], dtype=np.int64))
            off.append(off[-1] + len(starts[-1]))
    cat = lambda x: np.concatenate(x) if x else np.zeros(0, dtype=np.int64)
    return dict(cls=np.array(cls, dtype=np.int64), lab=np.array(lab, dtype=np.int64),
                box=np.array(box, dtype=np.int64).reshape(len(lab), -1), off=np.array(off, dtype=np.int64),
                starts=cat(starts), runs=cat(runs))


def _pack_instances(instances, prefix):
    d = _pack_rle_seg({0: instances})
    return {f'{prefix}_{k}': v for k, v in d.items() if k != 'cls'}


def main():
    assert os.path.isdir(REF), "reference not mounted: goldens can only be generated in the build container"
    _install_standins()
    import torch
    from empanada import array_utils as AU
    from empanada import consensus as CO
    from empanada.inference import engines as EN
    from empanada.inference import filters as FI
    from empanada.inference import matcher as MA
    from empanada.inference import patterns as PA
    from empanada.inference import postprocess as PP
    from empanada.inference import rle as RL
    from empanada.inference import tracker as TR

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    from empanada_amd import synthetic as SY

    rng = np.random.default_rng(20261004)

    # ------------------------------------------------------------------ P3 centres
    cases = {}
    for i, (h, w, thr, k) in enumerate([(64, 80, 0.1, 7), (50, 50, 0.1, 3), (40, 64, 0.3, 4),
                                        (33, 47, 0.05, 5), (32, 32, 0.1, 1), (64, 64, 0.1, 6)]):
        hm = rng.random((h, w), dtype=np.float32) ** 6
        # plateaus and exact-threshold values
        hm[5:7, 5:8] = 0.5
        hm[10, 10] = np.float32(thr)
        hm[12, 12] = np.nextafter(np.float32(thr), np.float32(1))
        hm[0, 0] = 0.9
        hm[h - 1, w - 1] = 0.95
        ctr = PP.find_instance_center(torch.from_numpy(hm.copy())[None, None], float(thr), int(k)).numpy()
        cases[f'c{i}_hmp'] = hm
        cases[f'c{i}_par'] = np.array([thr, k], dtype=np.float64)
        cases[f'c{i}_ctr'] = ctr
    _save('find_centers', n=np.array(6), **cases)

    # ------------------------------------------------------------------ P4 grouping
    cases = {}
    specs = [(48, 64, 5, 1), (48, 64, 20, 1), (48, 64, 21, 1), (40, 40, 57, 1), (16, 24, 3, 4),
             (16, 24, 45, 4), (32, 32, 2, 1), (24, 24, 41, 1)]
    for i, (h, w, K, step) in enumerate(specs):
        ctr = np.stack([rng.integers(0, h, K), rng.integers(0, w, K)], axis=1).astype(np.int64)
        off = (rng.normal(0, 6 * step, (1, 2, h, w))).astype(np.float32)
        if i == 6:      # adversarial exact ties: zero offsets, symmetric centres
            ctr = np.array([[10, 10], [10, 20]], dtype=np.int64)
            off[:] = 0
        if i == 7:      # everything farther than 1e5 for K > 20 -> id 0 rows, plus near ties
            off[0, :, :4] = 3e5
            off[0, :, 4:] = np.round(off[0, :, 4:])
        ids = PP.group_pixels(torch.from_numpy(ctr), torch.from_numpy(off), step=float(step)).numpy()
        cases[f'g{i}_ctr'] = ctr
        cases[f'g{i}_off'] = off
        cases[f'g{i}_step'] = np.array(step)
        cases[f'g{i}_ids'] = ids
    _save('group_pixels', n=np.array(len(specs)), **cases)

    # ------------------------------------------------------------------ P5 merge semantic + instance
    cases = {}
    for i, (h, w, C, thing, div, stuff, void) in enumerate([
            (40, 48, 1, [1], 1000, 64, 0), (40, 48, 4, [1, 3], 1000, 50, 0),
            (32, 32, 3, [2], 20000, 10, 0), (32, 40, 4, [1, 2], 1000, 2000, 255)]):
        sem = rng.integers(0, C + 1, (1, h, w)).astype(np.int64)
        sem = np.repeat(np.repeat(sem[:, ::4, ::4], 4, 1), 4, 2)      # blocky
        ins = rng.integers(0, 9, (1, h // 8, w // 8)).astype(np.int64)
        ins = np.repeat(np.repeat(ins, 8, 1), 8, 2)
        ins = ins * np.isin(sem, thing)
        pan = PP.merge_semantic_and_instance(torch.from_numpy(sem), torch.from_numpy(ins), div, thing,
                                             stuff, void).numpy()
        cases.update({f'm{i}_sem': sem, f'm{i}_ins': ins, f'm{i}_pan': pan,
                      f'm{i}_par': np.array([div, stuff, void]), f'm{i}_thing': np.array(thing)})
    _save('merge_sem_ins', n=np.array(4), **cases)

    # ------------------------------------------------------------------ P1 median queue
    cases = {}
    for i, (ks, n) in enumerate([(3, 6), (5, 9), (7, 12), (1, 4), (5, 3), (3, 2), (7, 7)]):
        xs = rng.random((n, 1, 2, 6, 7), dtype=np.float32)
        q = EN._MedianQueue(ks)
        outs, emitted = [], []
        for t in range(n):
            q.enqueue({'sem': torch.from_numpy(xs[t].copy()), 't': t})
            o = q.get_next(['sem'])
            if o is not None:
                outs.append(o['sem'].numpy().copy())
                emitted.append(o['t'])
        for o in q.end():
            outs.append(o['sem'].numpy().copy())
            emitted.append(o['t'])
        cases.update({f'q{i}_x': xs, f'q{i}_ks': np.array(ks), f'q{i}_out': np.stack(outs) if outs else np.zeros(0),
                      f'q{i}_emitted': np.array(emitted)})
    _save('median_queue', n=np.array(7), **cases)

    # ------------------------------------------------------------------ engines on planted heads
    class Stub(torch.nn.Module):
        """returns pre-computed head tensors slice by slice (logits = logit(prob) is avoided:
        the engine is patched to take probabilities directly)"""
        def __init__(self, heads):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.heads, self.t = heads, 0

        def forward(self, x, *a, **k):
            o = {k2: v[self.t:self.t + 1].clone() for k2, v in self.heads.items()}
            o['sem_logits'] = o.pop('sem')
            self.t += 1
            return o

    _orig = EN.logits_to_prob
    EN.logits_to_prob = lambda x: x          # stub already yields probabilities
    cases = {}
    ecfgs = [
        dict(shape=(10, 64, 72), axis='xy', ks=3, coarse=False, render=True, C=1, thr=0.3, nk=7),
        dict(shape=(12, 64, 64), axis='xz', ks=5, coarse=True, render=True, C=1, thr=0.5, nk=3),
        dict(shape=(9, 48, 80), axis='xy', ks=3, coarse=False, render=False, C=1, thr=0.5, nk=7),
        dict(shape=(10, 64, 64), axis='yz', ks=3, coarse=False, render=True, C=3, thr=0.5, nk=7),
        dict(shape=(8, 64, 64), axis='xy', ks=7, coarse=True, render=True, C=2, thr=0.5, nk=7),
    ]
    for i, e in enumerate(ecfgs):
        lab, cls = SY.planted_labels(e['shape'], fill=0.15, rmin=4, rmax=10, seed=100 + i, n_classes=e['C'])
        heads = SY.planted_heads(lab, cls, e['axis'], n_classes=e['C'], seed=7 + i, coarse=e['coarse'])
        thing = [1] if e['C'] == 1 else list(range(1, e['C']))      # last class = stuff when C > 1
        S, _, H, W = heads['sem'].shape
        kw = dict(thing_list=thing, label_divisor=1000, stuff_area=32, void_label=0, nms_threshold=0.1,
                  nms_kernel=e['nk'], confidence_thr=e['thr'], median_kernel_size=e['ks'])
        stub = Stub(heads)
        outs = []
        if e['render']:
            eng = EN.PanopticDeepLabRenderEngine3d(stub, padding_factor=16, coarse_boundaries=e['coarse'], **kw)
            for t in range(S):
                o = eng(torch.zeros(1, 1, H, W), (H - 3, W - 5))
                if o is not None:
                    outs.append(o.numpy())
            outs += [o.numpy() for o in eng.end()]
        else:
            eng = EN.PanopticDeepLabEngine3d(stub, **kw)
            for t in range(S):
                o = eng(torch.zeros(1, 1, H, W))
                if o is not None:
                    outs.append(o.numpy())
            outs += [o.numpy() for o in eng.end()]
        cases.update({f'e{i}_sem': heads['sem'].numpy(), f'e{i}_ctr': heads['ctr_hmp'].numpy(),
                      f'e{i}_off': heads['offsets'].numpy(), f'e{i}_pan': np.stack(outs),
                      f'e{i}_thing': np.array(thing),
                      f'e{i}_par': np.array([e['ks'], int(e['coarse']), int(e['render']), e['C'], e['nk']]),
                      f'e{i}_thr': np.array(e['thr'])})
    EN.logits_to_prob = _orig
    _save('engines', n=np.array(len(ecfgs)), **cases)

    # ------------------------------------------------------------------ R2 pan_seg_to_rle_seg
    cases = {}
    for i, (h, w, fc) in enumerate([(40, 50, True), (40, 50, False), (64, 64, True), (17, 9, True)]):
        base = rng.integers(0, 4, (h // 2 + 1, w // 2 + 1))
        pan = np.repeat(np.repeat(base, 2, 0), 2, 1)[:h, :w].astype(np.int64)
        cls = rng.integers(1, 4, (h, w))
        pan = np.where(pan > 0, cls * 1000 + pan, 0).astype(np.int64)
        pan[rng.random((h, w)) < 0.1] = 3000          # stuff class 3, id 0
        rle = RL.pan_seg_to_rle_seg(pan, [1, 2, 3], 1000, [1, 2], fc)
        back = RL.rle_seg_to_pan_seg(rle, (h, w))
        cases.update({f'r{i}_pan': pan, f'r{i}_fc': np.array(fc), f'r{i}_back': back})
        cases.update({f'r{i}_{k}': v for k, v in _pack_rle_seg(rle).items()})
    _save('rle_seg', n=np.array(4), **cases)

    # ------------------------------------------------------------------ array_utils properties (seeded)
    cases = {}
    for i in range(6):
        n = 4000
        a = np.unique(rng.integers(0, n, rng.integers(1, 900)))
        b = np.unique(rng.integers(0, n, rng.integers(1, 900)))
        c = np.unique(rng.integers(0, n, rng.integers(1, 900)))
        sa, ra = AU.rle_encode(a)
        sb, rb = AU.rle_encode(b)
        sc, rc = AU.rle_encode(c)
        rngs = [np.stack([s, s + r], 1) for s, r in ((sa, ra), (sb, rb), (sc, rc))]
        cases.update({
            f'u{i}_a': a, f'u{i}_b': b, f'u{i}_c': c, f'u{i}_sa': sa, f'u{i}_ra': ra,
            f'u{i}_inter': np.array(AU.rle_intersection(sa, ra, sb, rb)),
            f'u{i}_iou': np.array(AU.rle_iou(sa, ra, sb, rb)),
            f'u{i}_ioa': np.array(AU.rle_ioa(sa, ra, sb, rb)),
            f'u{i}_vote2': np.array(AU.vote_by_ranges([r.copy() for r in rngs], 2)),
            f'u{i}_vote3': np.array(AU.vote_by_ranges([r.copy() for r in rngs], 3)),
            f'u{i}_join': np.array(AU.vote_by_ranges([r.copy() for r in rngs], 1)),
        })
        ms, mr = AU.merge_rles(sa, ra, sb, rb)
        cases.update({f'u{i}_ms': ms, f'u{i}_mr': mr})
    # malformed (overlapping / unsorted) rles exercise the literal sweep
    for i in range(6, 10):
        sa = rng.integers(0, 300, 40); ra = rng.integers(1, 30, 40)
        sb = rng.integers(0, 300, 35); rb = rng.integers(1, 30, 35)
        cases.update({f'u{i}_sa': sa, f'u{i}_ra': ra, f'u{i}_sb': sb, f'u{i}_rb': rb,
                      f'u{i}_inter': np.array(AU.rle_intersection(sa, ra, sb, rb))})
    # boxes
    b1 = rng.integers(0, 60, (30, 3)); b1 = np.concatenate([b1, b1 + rng.integers(1, 25, (30, 3))], 1)
    b2 = rng.integers(0, 60, (25, 3)); b2 = np.concatenate([b2, b2 + rng.integers(1, 25, (25, 3))], 1)
    iou, inter = AU.box_iou(b1, b2, return_intersection=True)
    cases.update({'box3_a': b1, 'box3_b': b2, 'box3_iou': iou.toarray(), 'box3_inter': inter.toarray()})
    b1 = b1[:, [0, 1, 3, 4]]; b2 = b2[:, [0, 1, 3, 4]]
    iou, inter = AU.box_iou(b1, b2, return_intersection=True)
    cases.update({'box2_a': b1, 'box2_b': b2, 'box2_iou': iou.toarray(), 'box2_inter': inter.toarray()})
    _save('array_utils', **cases)

    # ------------------------------------------------------------------ M4/M5 matcher chain + T1 trackers
    # the reference's own KAT (tests/test_matcher.py:6-66), fixture values re-expressed as data
    tgt = np.zeros((200, 200), np.uint32); mat = np.zeros((200, 200), np.uint32); exp = np.zeros((200, 200), np.uint32)
    for arr, boxes in ((tgt, [((0, 16, 0, 16), 1001), ((30, 50, 30, 50), 1002), ((0, 10, 190, 200), 1003),
                              ((150, 200, 0, 50), 1004), ((170, 200, 170, 200), 1005), ((100, 130, 90, 110), 1006)]),
                       (mat, [((0, 16, 0, 16), 1009), ((30, 50, 30, 50), 1008), ((0, 10, 190, 200), 1007),
                              ((150, 200, 0, 50), 1006), ((170, 200, 45, 80), 1005), ((180, 200, 180, 200), 1004),
                              ((100, 115, 90, 110), 1003), ((115, 130, 90, 110), 1002), ((50, 75, 125, 160), 1001)]),
                       (exp, [((0, 16, 0, 16), 1001), ((30, 50, 30, 50), 1002), ((0, 10, 190, 200), 1003),
                              ((150, 200, 0, 50), 1004), ((170, 200, 45, 80), 1008), ((180, 200, 180, 200), 1005),
                              ((100, 115, 90, 110), 1006), ((115, 130, 90, 110), 1006), ((50, 75, 125, 160), 1007)])):
        for (y0, y1, x0, x1), v in boxes:
            arr[y0:y1, x0:x1] = v
    m = MA.RLEMatcher(1, 1000, 0.25, 0.25, True)
    trle = RL.pan_seg_to_rle_seg(tgt, [1], 1000, [1], False)
    mrle = RL.pan_seg_to_rle_seg(mat, [1], 1000, [1], False)
    m.initialize_target(trle[1])
    mrle[1] = m(mrle[1], update_target=False)
    got = RL.rle_seg_to_pan_seg(mrle, (200, 200))
    assert np.array_equal(got, exp), "reference failed its own test_matcher KAT under the stand-ins"
    _save('matcher_kat', target=tgt, match=mat, out=exp)

    # full forward/backward chain through planted stacks, all three axes, then consensus
    cases = {}
    pcfgs = [dict(shape=(40, 48, 56), C=1, ks=3, seed=11), dict(shape=(36, 40, 44), C=3, ks=5, seed=12)]
    for i, pc in enumerate(pcfgs):
        shape = pc['shape']
        lab, cls = SY.planted_labels(shape, fill=0.18, rmin=4, rmax=10, seed=pc['seed'], n_classes=pc['C'])
        C = pc['C']
        thing = [1] if C == 1 else list(range(1, C))
        labels = [1] if C == 1 else list(range(1, C + 1))
        div = 1000
        trackers = PA.create_axis_trackers({'xy': 0, 'xz': 1, 'yz': 2}, labels, div, shape)
        EN.logits_to_prob = lambda x: x
        for axis_name, ax in (('xy', 0), ('xz', 1), ('yz', 2)):
            heads = SY.planted_heads(lab, cls, axis_name, n_classes=C, seed=31 + i, coarse=False)
            S, _, H, W = heads['sem'].shape
            eng = EN.PanopticDeepLabRenderEngine3d(
                Stub(heads), thing_list=thing, label_divisor=div, stuff_area=16, void_label=0,
                nms_threshold=0.1, nms_kernel=7, confidence_thr=0.5, median_kernel_size=pc['ks'],
                padding_factor=8, coarse_boundaries=False)
            pans = []
            for t in range(S):
                o = eng(torch.zeros(1, 1, H, W), (H, W))
                if o is not None:
                    pans.append(o.squeeze().numpy())
            pans += [o.squeeze().numpy() for o in eng.end()]
            assert len(pans) == S
            matchers = PA.create_matchers(thing, div, 0.25, 0.25)
            rle_stack = []
            for pan in pans:
                rs = RL.pan_seg_to_rle_seg(pan, labels, div, thing, force_connected=True)
                rle_stack.append(PA.apply_matchers(rs, matchers))
            fwd = np.stack([RL.rle_seg_to_pan_seg(rs, pans[0].shape) for rs in rle_stack])
            for idx, rs in PA.backward_matching(rle_stack, matchers, S):
                PA.update_trackers(rs, idx, trackers[axis_name])
            bwd = np.stack([RL.rle_seg_to_pan_seg(rs, pans[0].shape) for rs in rle_stack])
            PA.finish_tracking(trackers[axis_name])
            # heads are regenerated by the tests from (labels, seed) with the committed generator
            cases.update({f'p{i}_{axis_name}_pan': np.stack(pans).astype(np.int32),
                          f'p{i}_{axis_name}_fwd': fwd, f'p{i}_{axis_name}_bwd': bwd,
                          f'p{i}_{axis_name}_semsum': np.array(float(heads['sem'].double().sum()))})
            for tr in trackers[axis_name]:
                cases.update(_pack_instances(tr.instances, f'p{i}_{axis_name}_tr{tr.class_id}'))
                FI.remove_small_objects(tr, min_size=100)
                FI.remove_pancakes(tr, min_span=3)
        EN.logits_to_prob = _orig
        for cid in labels:
            cts = PA.get_axis_trackers_by_class(trackers, cid)
            if cid in thing:
                con = PA.create_instance_consensus(cts, 2, 0.75, False)
                FI.remove_small_objects(con, min_size=100)
                FI.remove_pancakes(con, min_span=3)
            else:
                con = PA.create_semantic_consensus(cts, 2)
            vol = np.zeros(shape, dtype=np.uint32)
            PA.fill_volume(vol, con.instances)
            cases.update(_pack_instances(con.instances, f'p{i}_con{cid}'))
            cases[f'p{i}_vol{cid}'] = vol
        cases[f'p{i}_lab'] = lab
        cases[f'p{i}_cls'] = cls
        cases[f'p{i}_par'] = np.array([C, pc['ks'], pc['seed'], 31 + i])
    _save('pipeline', n=np.array(len(pcfgs)), **cases)

    # ------------------------------------------------------------------ C4 consensus KATs (tests/test_consensus.py)
    s2 = SY.ball(20).astype(np.uint32)
    s4 = s2.copy(); s4[:, 20:, 20:] = 0
    shape = (100, 100, 100)
    vols = [np.zeros(shape, np.uint32) for _ in range(3)]
    vols[0][:41, :41, :41][s2 > 0] = 1001
    vols[0][15:56, 15:56, 15:56][s2 > 0] = 1002
    vols[1][:41, :41, :41][s2 > 0] = 1005
    vols[1][15:56, 15:56, 15:56][s4 > 0] = 1004
    vols[1][:41, 59:100, 59:100][s2 > 0] = 1006
    vols[2][:41, :41, :41][s2 > 0] = 1003
    vols[2][15:56, 15:56, 15:56][s4 > 0] = 1003
    trs = [TR.InstanceTracker(1, 1000, shape, axis='xy') for _ in range(3)]
    for z in range(100):
        for v, tr in zip(vols, trs):
            tr.update(RL.pan_seg_to_rle_seg(v[z], [1], 1000, [1], force_connected=False)[1], z)
    for tr in trs:
        tr.finish()
    cases = {'vols': np.stack(vols)}
    for j, (vote, iou_thr, bypass) in enumerate([(2, 0.75, False), (2, 0.75, True), (1, 0.75, False),
                                                 (3, 0.75, False), (2, 0.1, False), (1, 0.75, True)]):
        inst = CO.merge_objects_from_trackers(trs, vote, iou_thr, bypass)
        vol = np.zeros(shape, np.uint32)
        AU.numpy_fill_instances(vol, inst)
        # stored as RLE of the flat volume to keep the fixture small
        flat = vol.ravel()
        ch = np.flatnonzero(np.diff(flat)) + 1
        st = np.concatenate([[0], ch]); ln = np.diff(np.concatenate([st, [flat.size]]))
        cases.update({f'k{j}_par': np.array([vote, iou_thr, int(bypass)]), f'k{j}_st': st, f'k{j}_ln': ln,
                      f'k{j}_val': flat[st]})
        cases.update(_pack_instances(inst, f'k{j}_inst'))
    for j, vote in enumerate([2, 1]):
        # semantic: one instance per tracker
        strs = []
        for v in vols:
            tr = TR.InstanceTracker(1, 1000, shape, axis='xy')
            for z in range(100):
                tr.update(RL.pan_seg_to_rle_seg((v[z] > 0).astype(np.uint32) * 1000, [1], 1000, [], False)[1], z)
            tr.finish()
            strs.append(tr)
        inst = CO.merge_semantic_from_trackers(strs, vote)
        cases.update(_pack_instances(inst, f's{j}_inst'))
        cases[f's{j}_vote'] = np.array(vote)
    _save('consensus_kat', **cases)

    # ------------------------------------------------------------------ T1 trackers incl. xz wrap bug, Z1 chunk_ranges
    vol = rng.integers(0, 6, size=(24, 20, 28)).astype(np.uint32)
    vol[vol > 0] += 1000
    cases = {'vol': vol}
    for axis_name, ax in (('xy', 0), ('xz', 1), ('yz', 2)):
        tr = TR.InstanceTracker(1, 1000, vol.shape, axis=axis_name)
        for idx in range(vol.shape[ax]):
            sl = np.take(vol, idx, axis=ax)
            tr.update(RL.pan_seg_to_rle_seg(sl, [1], 1000, [1], False)[1], idx)
        tr.finish()
        filled = AU.numpy_fill_instances(np.zeros_like(vol), tr.instances)
        cases.update(_pack_instances(tr.instances, f't_{axis_name}'))
        cases[f't_{axis_name}_filled'] = filled
    from empanada import zarr_utils as ZU
    rr = np.stack([rng.integers(0, 24 * 20 * 28 - 200, 50), rng.integers(1, 200, 50)], 1)
    rr = np.cumsum(rr, axis=1)
    cr = np.array(ZU.chunk_ranges(rr, 24 * 20 * 28, 5 * 20 * 28))
    cr2 = np.array(ZU.chunk_ranges(cr, 20 * 28, 7 * 28))
    cr3 = np.array(ZU.chunk_ranges(cr2, 28, 10))
    cases.update({'z_ranges': rr, 'z_c1': cr, 'z_c2': cr2, 'z_c3': cr3, 'z_chunks': np.array([5, 7, 10])})
    # Z1 zarr_fill_instances on a numpy-backed array with the zarr.Array attributes it reads; the process pool
    # is run serially (an in-memory array cannot be shared with worker processes)
    class _Arr:
        def __init__(self, shape, chunks):
            import math
            self.a = np.zeros(shape, np.uint32); self.shape = shape; self.chunks = chunks
            self.nchunks = math.prod(math.ceil(s / c) for s, c in zip(shape, chunks))
        def __getitem__(self, i): return self.a[i]
        def __setitem__(self, i, v): self.a[i] = v
    class _SerialPool:
        def __init__(self, n): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def map(self, f, it): return [f(x) for x in it]
    ZU.Pool = _SerialPool
    tr = TR.InstanceTracker(1, 1000, vol.shape, axis='xy')
    for idx in range(vol.shape[0]):
        tr.update(RL.pan_seg_to_rle_seg(vol[idx], [1], 1000, [1], False)[1], idx)
    tr.finish()
    zchunks = [(22, 15, 17), (10, 9, 5), (6, 5, 9), (21, 15, 26), (24, 20, 28), (1, 20, 28), (7, 20, 27)]
    for j, ch in enumerate(zchunks):
        arr = _Arr(vol.shape, ch)
        ZU.zarr_fill_instances(arr, tr.instances, 4)
        cases[f'zf{j}_chunks'] = np.array(ch)
        cases[f'zf{j}_vol'] = arr.a
    cases['zf_n'] = np.array(len(zchunks))
    _save('trackers', **cases)


    # ------------------------------------------------------------------ C5 tiles
    # cztile is absent; empanada/inference/tile.py imports it at module level.  A stub satisfies the import and a
    # Tiler is assembled with THIS repository's documented geometry, so that the reference's own
    # calculate_overlap_rle / translate_rle_seg / merge_objects_from_tiles produce the expected values.
    cz = types.ModuleType('cztile')
    cz1 = types.ModuleType('cztile.fixed_total_area_strategy'); cz1.AlmostEqualBorderFixedTotalAreaStrategy2D = object
    cz2 = types.ModuleType('cztile.tiling_strategy'); cz2.Rectangle = object
    sys.modules.update({'cztile': cz, 'cztile.fixed_total_area_strategy': cz1, 'cztile.tiling_strategy': cz2})
    from empanada.inference import tile as TL
    from empanada_amd.inference.tile import axis_offsets
    shape = (400, 400)
    rr = np.arange(-20, 21)
    circle = ((rr[:, None] ** 2 + rr[None, :] ** 2) <= 400).astype(np.uint32)      # skimage.morphology.disk(20)
    seg = np.zeros(shape, dtype=np.uint32)
    c = 1001
    for xs in range(0, 351, 50):
        for ys in range(0, 351, 50):
            seg[ys:ys + 41, xs:xs + 41][circle > 0] = c
            c += 1
    seg[380:, 100:300] = 2000                                                        # a semantic (stuff) band
    cases = {'seg': seg}
    for ti, (tsize, ov) in enumerate([((100, 110), 20), ((256, 256), 64)]):
        yr, xr = [], []
        for y in axis_offsets(shape[0], tsize[0], ov):
            for x in axis_offsets(shape[1], tsize[1], ov):
                yr.append((y, y + min(tsize[0], shape[0]))); xr.append((x, x + min(tsize[1], shape[1])))
        tl = TL.Tiler.__new__(TL.Tiler)
        tl.image_shape, tl.yranges, tl.xranges = shape, yr, xr
        ovs, ovr = TL.calculate_overlap_rle(yr, xr, shape)
        thing_tiles, stuff_tiles = [], []
        for i in range(len(yr)):
            crop = seg[yr[i][0]:yr[i][1], xr[i][0]:xr[i][1]]
            lab = sys.modules['skimage.measure'].label(np.where(crop < 2000, crop, 0)).astype(np.uint32)
            lab[lab > 0] += 1000
            lab[crop == 2000] = 2000
            rs = tl.translate_rle_seg(RL.pan_seg_to_rle_seg(lab, [1, 2], 1000, [1], False), i)
            thing_tiles.append(rs[1]); stuff_tiles.append(rs[2])
        cp = lambda tiles: [{k: {kk: (vv.copy() if hasattr(vv, 'copy') else vv) for kk, vv in v.items()} for k, v in t.items()} for t in tiles]
        merged = CO.merge_objects_from_tiles(cp(thing_tiles))
        merged_ov = CO.merge_objects_from_tiles(cp(thing_tiles), (ovs, ovr))
        sem = CO.merge_semantic_from_tiles(cp(stuff_tiles))
        cases.update({f't{ti}_par': np.array([tsize[0], tsize[1], ov]), f't{ti}_yr': np.array(yr), f't{ti}_xr': np.array(xr),
                      f't{ti}_ovs': np.asarray(ovs), f't{ti}_ovr': np.asarray(ovr)})
        cases.update(_pack_instances(merged, f't{ti}_merged'))
        cases.update(_pack_instances(merged_ov, f't{ti}_mergedov'))
        cases.update(_pack_instances(sem, f't{ti}_sem'))
    _save('tiles', n=np.array(2), **cases)

    # ------------------------------------------------------------------ D1 model forwards (reference classes)
    from empanada.models.panoptic_deeplab import PanopticDeepLab as RefPDL
    from empanada.models.quantization.panoptic_deeplab import QuantizablePanopticDeepLabPR as RefQPR
    from empanada.models.panoptic_bifpn import PanopticBiFPN as RefBiFPN
    from empanada.models.quantization.panoptic_bifpn import QuantizablePanopticBiFPNPR as RefQBiFPNPR
    from empanada_amd.models import PanopticBiFPN, PanopticBiFPNPR, PanopticDeepLab, PanopticDeepLabPR, synthesize_weights
    cases = {}
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 1, 128, 128, generator=g)
    cases['x'] = x.numpy()
    mito = dict(encoder='resnet50', num_classes=1, stage4_stride=16, decoder_channels=256, low_level_stages=[1],
                low_level_channels_project=[32], atrous_rates=[2, 4, 6], aspp_channels=None, aspp_dropout=0.5,
                ins_decoder=True, ins_ratio=0.5)
    for name, ours, ref, args in (
            ('pdl_r50', PanopticDeepLab(encoder='resnet50', num_classes=1), RefPDL(encoder='resnet50', num_classes=1), ()),
            ('pdl_r50_c5', PanopticDeepLab(encoder='resnet50', num_classes=5), RefPDL(encoder='resnet50', num_classes=5), ()),
            ('pdlpr_mito', PanopticDeepLabPR(**mito), RefQPR(quantize=False, **mito), (3, False)),
            ('bifpn_regnety', PanopticBiFPN(encoder='regnety_6p4gf', num_classes=1),
             RefBiFPN(encoder='regnety_6p4gf', num_classes=1), ()),
            ('bifpnpr_r50', PanopticBiFPNPR(encoder='resnet50', num_classes=3, ins_decoder=True),
             RefQBiFPNPR(quantize=False, encoder='resnet50', num_classes=3, ins_decoder=True), (2, True))):
        synthesize_weights(ours)            # weights are a function of the state-dict keys only
        with torch.no_grad():
            for head in (ours.semantic_head, ours.ins_center, ours.ins_xy):
                head.head[1].weight.mul_(1e-3)
        ref.load_state_dict(ours.state_dict(), strict=True)
        ref.eval()
        with torch.no_grad():
            out = ref(x, *args)
        for k in ('sem_logits', 'ctr_hmp', 'offsets'):
            cases[f'{name}_{k}'] = out[k].numpy()
    _save('models', **cases)


if __name__ == '__main__':
    main()
