"""Whole-volume drivers of the CPU oracle (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

The call sequence of scripts/pdl_inference3d.py:140-233 (engine per slice -> pan_seg_to_rle_seg -> forward /
backward matching -> trackers -> filters -> instance consensus -> filters -> fill) over the oracle's own functions,
in one place for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

`workers > 1` spreads the two per-pixel stages over fresh (spawned, numpy-only: they never see the GPU) processes so
that BASELINE.json's full-size configurations finish in about a minute on the GPU box's host cores:
  * the recursive median is element-wise in (y, x): every worker drives the oracle's MedianQueue over ALL slices of its
    own strip of rows (the recursion in z stays serial inside a strip);
  * everything from the filtered probabilities to the slice's rle_seg is independent per slice.
The sequential stages (matchers, trackers, consensus) run in the calling process, in the reference's order.  The
functions called are exactly the serial oracle's (postprocess.post_slice, rle_seg.pan_seg_to_rle_seg, ...), only the
loop over slices is distributed; tests/test_oracle_pipeline.py holds workers=3 to workers=1 bit for bit.
"""
import os
import shutil
import tempfile

import numpy as np

from . import consensus as OC
from . import postprocess as OP
from . import rle_ops as OR
from . import rle_seg as OS

AXES = ('xy', 'xz', 'yz')


def _scratch():
    base = '/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir()
    return tempfile.mkdtemp(prefix='emp_oracle_', dir=base)


def _median_strip(job):
    """worker: recursive median (engines.py:68-90) over all slices for rows [r0, r1) -> sem_filt[:, :, r0:r1]"""
    d, ks, r0, r1 = job
    sem = np.load(os.path.join(d, 'sem.npy'), mmap_mode='r')
    out = np.load(os.path.join(d, 'sem_filt.npy'), mmap_mode='r+')
    q = OP.MedianQueue(ks)

    def emit(o):
        out[o['t'], :, r0:r1] = o['sem'][0]

    for t in range(sem.shape[0]):
        q.enqueue({'sem': np.array(sem[t:t + 1, :, r0:r1], dtype=np.float32), 't': t})
        o = q.get_next(['sem'])
        if o is not None:
            emit(o)
    for o in q.end():
        emit(o)
    out.flush()
    return r0


def _slices(job):
    """worker: filtered probabilities -> pan (postprocess.post_slice) -> rle_seg, for the listed slices"""
    d, ts, kw, labels, render, sizes = job
    sem = np.load(os.path.join(d, 'sem_filt.npy'), mmap_mode='r')
    ctr = np.load(os.path.join(d, 'ctr.npy'), mmap_mode='r')
    off = np.load(os.path.join(d, 'off.npy'), mmap_mode='r')
    pan_out = np.load(os.path.join(d, 'pan.npy'), mmap_mode='r+')
    res = []
    for t in ts:
        size = tuple(sizes[t]) if sizes is not None else tuple(sem.shape[-2:])
        o = {'sem': np.array(sem[t:t + 1]), 'ctr_hmp': np.array(ctr[t:t + 1]), 'offsets': np.array(off[t:t + 1]),
             'size': size}
        pan = np.asarray(OP.post_slice(o, render=render, **kw)).squeeze()
        pan_out[t, :pan.shape[0], :pan.shape[1]] = pan
        res.append((t, OS.pan_seg_to_rle_seg(pan, labels, kw['label_divisor'], kw['thing_list'], force_connected=True)))
    pan_out.flush()
    return res


def emitted_slices(n, ks):
    """indices of the slices a 3d engine emits for a stack of n slices, in emission order (engines.py:68-90: a stack
    shorter than the kernel loses its tail)"""
    q = OP.MedianQueue(ks)
    order = []
    for t in range(n):
        q.enqueue({'t': t})
        o = q.get_next([])
        if o is not None:
            order.append(o['t'])
    order += [o['t'] for o in q.end()]
    return order


def plane_pans(sem, ctr, off, engine, *, labels, render=True, sizes=None, workers=1, with_rle=True):
    """One plane's stack of head tensors (numpy: sem (n,C,H,W), ctr (n,1,h,w), off (n,2,h,w)) -> (pans, rle_segs) in
    emission order: pans[i] (H,W) int64 as the engine returns them, rle_segs[i] = pan_seg_to_rle_seg(pans[i])."""
    kw = dict(engine)
    ks = kw.pop('median_kernel_size')
    kw.setdefault('coarse_boundaries', False)
    n = sem.shape[0]
    if workers <= 1:
        pans = OP.engine3d_stack([sem[t:t + 1] for t in range(n)], [ctr[t:t + 1] for t in range(n)],
                                 [off[t:t + 1] for t in range(n)], render=render, sizes=sizes,
                                 median_kernel_size=ks, **kw)
        pans = [np.asarray(p).squeeze() for p in pans]
        rles = [OS.pan_seg_to_rle_seg(p, labels, kw['label_divisor'], kw['thing_list'], force_connected=True)
                for p in pans] if with_rle else None
        return pans, rles
    order = emitted_slices(n, ks)
    d = _scratch()
    try:
        np.save(os.path.join(d, 'sem.npy'), np.ascontiguousarray(sem, dtype=np.float32))
        np.save(os.path.join(d, 'ctr.npy'), np.ascontiguousarray(ctr, dtype=np.float32))
        np.save(os.path.join(d, 'off.npy'), np.ascontiguousarray(off, dtype=np.float32))
        H, W = sem.shape[-2:]
        np.lib.format.open_memmap(os.path.join(d, 'sem_filt.npy'), mode='w+', dtype=np.float32, shape=sem.shape).flush()
        np.lib.format.open_memmap(os.path.join(d, 'pan.npy'), mode='w+', dtype=np.int64, shape=(n, H, W)).flush()
        rows = np.linspace(0, H, min(workers, H) + 1).astype(int)
        _run_workers(d, 'median', [(d, ks, int(a), int(b)) for a, b in zip(rows[:-1], rows[1:]) if b > a], workers)
        # interleaved slices per worker: object load varies smoothly along the axis
        jobs = [(d, order[i::workers], kw, list(labels), render, sizes) for i in range(workers) if order[i::workers]]
        got = _run_workers(d, 'slices', jobs, workers)
        rle_by_t = {t: r for part in got for t, r in part}
        pan = np.load(os.path.join(d, 'pan.npy'), mmap_mode='r')
        pans = []
        for t in order:
            h, w = sizes[t] if (sizes is not None and render) else (H, W)
            pans.append(np.array(pan[t, :h, :w]))
        return pans, [rle_by_t[t] for t in order]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _run_workers(d, kind, jobs, workers):
    """one fresh interpreter per job (`python -m oracle.pipeline <kind> <job file> <result file>`): numpy-only children
    that depend neither on the caller's __main__ nor on its GPU state"""
    import pickle
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for i, job in enumerate(jobs):
        jf, rf = os.path.join(d, f'{kind}_{i}.job'), os.path.join(d, f'{kind}_{i}.res')
        with open(jf, 'wb') as f:
            pickle.dump(job, f)
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        procs.append((subprocess.Popen([sys.executable, '-m', 'oracle.pipeline', kind, jf, rf], cwd=root, env=env), rf))
    out = []
    failed = [p.args for p, _ in procs if p.wait() != 0]
    if failed:
        raise RuntimeError(f'oracle worker failed: {failed[0]}')
    for _, rf in procs:
        with open(rf, 'rb') as f:                      # files this module wrote itself a moment ago
            out.append(pickle.load(f))
    return out


def tiled_plane_pans(sem, ctr, off, engine, yranges, xranges, overlap_rle, *, labels, single_run='raise', workers=1):
    """The reference's tiled sequence (tests/test_tiling.py:26-47) over a stack: every tile's crop of the heads through
    the 3d engine (the recursive median runs along z inside each tile position) -> pan_seg_to_rle_seg(force_connected
    = False) -> Tiler.translate_rle_seg (inference/tile.py:122-168) -> merge_objects_from_tiles with the overlap test /
    merge_semantic_from_tiles per class (consensus.py:471-625) -> painted panoptic image of the whole plane, per slice.
    yranges / xranges / overlap_rle: the tiler's.  Returns the list of (H, W) uint32 images."""
    H, W = sem.shape[-2:]
    kw = dict(engine)
    things, div = kw['thing_list'], kw['label_divisor']
    per_tile = []
    for (y0, y1), (x0, x1) in zip(yranges, xranges):
        pans, _ = plane_pans(np.ascontiguousarray(sem[..., y0:y1, x0:x1]), np.ascontiguousarray(ctr[..., y0:y1, x0:x1]),
                             np.ascontiguousarray(off[..., y0:y1, x0:x1]), kw, labels=labels, workers=workers,
                             with_rle=False)
        per_tile.append([OS.pan_seg_to_rle_seg(p, labels, div, things, force_connected=False) for p in pans])
    out = []
    for z in range(len(per_tile[0])):
        img = np.zeros((H, W), dtype=np.uint32)
        for l in labels:
            moved = []
            for i, ((y0, y1), (x0, x1)) in enumerate(zip(yranges, xranges)):
                w = x1 - x0
                insts = {}
                for k, a in per_tile[i][z][l].items():
                    b = a['box']
                    insts[k] = {'box': (b[0] + y0, b[1] + x0, b[2] + y0, b[3] + x0), 'runs': a['runs'],
                                'starts': np.ravel_multi_index((a['starts'] // w + y0, a['starts'] % w + x0), (H, W))}
                moved.append(insts)
            merged = OC.merge_objects_from_tiles(moved, overlap_rle, single_run) if l in things else \
                OC.merge_semantic_from_tiles(moved, single_run)
            OR.numpy_fill_instances(img.reshape(-1), merged)
        out.append(img)
    return out


def plane_trackers(rle_segs, axis, shape3d, labels, thing_list, div, match, filters=None):
    """patterns.py:68-121 + tracker.py: per-slice rle_segs -> forward matching -> backward matching -> trackers
    (-> size / span filters).  `rle_segs` is consumed (matched in place like the reference's rle_stack)."""
    matchers = OS.create_matchers(thing_list, div, match['merge_iou_thr'], match['merge_ioa_thr'])
    stack = [OS.apply_matchers(r, matchers) for r in rle_segs]
    trackers = OS.create_axis_trackers([axis], labels, div, shape3d)[axis]
    for idx, rs in OS.backward_matching(stack, matchers, len(stack)):
        OS.update_trackers(rs, idx, trackers)
    OS.finish_tracking(trackers)
    if filters is not None:
        for tr in trackers:
            OS.remove_small_objects(tr, filters['min_size'])
            OS.remove_pancakes(tr, filters['min_span'])
    return trackers


def tracker_volume(trackers, shape3d):
    vol = np.zeros(shape3d, dtype=np.uint32)
    for tr in trackers:
        OR.numpy_fill_instances(vol, tr.instances)
    return vol


def stack_volume(heads, engine, match, filters, *, labels, workers=1):
    """BASELINE configs[1]: xy stack -> (pans, labelled uint32 volume of the tracked + filtered instances, #instances)"""
    sem, ctr, off = (np.asarray(heads[k]) for k in ('sem', 'ctr_hmp', 'offsets'))
    pans, rles = plane_pans(sem, ctr, off, engine, labels=labels, workers=workers)
    shape = (len(pans),) + tuple(pans[0].shape)
    trs = plane_trackers(rles, 'xy', shape, labels, engine['thing_list'], engine['label_divisor'], match, filters)
    return pans, tracker_volume(trs, shape), sum(len(t.instances) for t in trs)


def orthoplane_volume(heads, shape3d, engine, match, filters, consensus, *, labels, workers=1, timers=None):
    """BASELINE configs[2..3]: heads[axis] = {'sem','ctr_hmp','offsets'} (numpy) for xy / xz / yz -> ({class: labelled
    uint32 consensus volume}, #consensus instances, {axis: trackers}).  scripts/pdl_inference3d.py:140-233."""
    import time
    div, things = engine['label_divisor'], engine['thing_list']
    trackers = {}
    for axis in AXES:
        t0 = time.perf_counter()
        h = heads[axis]
        _, rles = plane_pans(np.asarray(h['sem']), np.asarray(h['ctr_hmp']), np.asarray(h['offsets']), engine,
                             labels=labels, workers=workers)
        t1 = time.perf_counter()
        trackers[axis] = plane_trackers(rles, axis, shape3d, labels, things, div, match, filters)
        if timers is not None:
            timers[f'{axis}_pixels_rle'] = t1 - t0
            timers[f'{axis}_match_track'] = time.perf_counter() - t1
    t0 = time.perf_counter()
    vols, n_inst = {}, 0
    for c in labels:
        cts = [t for axis in AXES for t in trackers[axis] if t.class_id == c]
        if c in things:
            con = OC.create_instance_consensus(cts, consensus['pixel_vote_thr'], consensus['cluster_iou_thr'],
                                               consensus['bypass'])
            OS.remove_small_objects(con, filters['min_size'])
            OS.remove_pancakes(con, filters['min_span'])
        else:
            con = OC.create_semantic_consensus(cts, consensus['pixel_vote_thr'])
        vols[c] = OR.numpy_fill_instances(np.zeros(shape3d, np.uint32), con.instances)
        n_inst += len(con.instances)
    if timers is not None:
        timers['consensus_fill'] = time.perf_counter() - t0
    return vols, n_inst, trackers


if __name__ == '__main__':
    import pickle
    import sys
    _kind, _jf, _rf = sys.argv[1:4]
    with open(_jf, 'rb') as _f:
        _job = pickle.load(_f)
    _res = {'median': _median_strip, 'slices': _slices}[_kind](_job)
    with open(_rf, 'wb') as _f:
        pickle.dump(_res, _f)
