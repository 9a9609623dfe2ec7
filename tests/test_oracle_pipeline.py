"""CPU: the whole oracle chain (engine -> RLE -> fwd/bwd matching -> trackers -> filters ->
consensus -> fill) against the chain run by the reference itself (tests/golden/pipeline.npz)."""
import numpy as np

from conftest import assert_instances_equal, load_golden, unpack_instances
from empanada_amd import synthetic as SY
from oracle import consensus as OC
from oracle import postprocess as OP
from oracle import rle_ops as OR
from oracle import rle_seg as OS


def run_oracle_pipeline(lab, cls, C, ks, head_seed, axes=('xy', 'xz', 'yz'), div=1000):
    shape = lab.shape
    thing = [1] if C == 1 else list(range(1, C))
    labels = [1] if C == 1 else list(range(1, C + 1))
    trackers = OS.create_axis_trackers(axes, labels, div, shape)
    per_axis = {}
    for name in axes:
        heads = SY.planted_heads(lab, cls, name, n_classes=C, seed=head_seed, coarse=False)
        sem, ctr, off = (heads[k].numpy() for k in ('sem', 'ctr_hmp', 'offsets'))
        S = sem.shape[0]
        pans = OP.engine3d_stack(
            [sem[t:t + 1] for t in range(S)], [ctr[t:t + 1] for t in range(S)], [off[t:t + 1] for t in range(S)],
            thing_list=thing, label_divisor=div, stuff_area=16, void_label=0, nms_threshold=0.1, nms_kernel=7,
            confidence_thr=0.5, median_kernel_size=ks, coarse_boundaries=False, render=True)
        pans = [p.squeeze() for p in pans]
        matchers = OS.create_matchers(thing, div, 0.25, 0.25)
        stack = OS.forward_matching(pans, matchers, labels, div, thing)
        fwd = np.stack([OS.rle_seg_to_pan_seg(rs, pans[0].shape) for rs in stack])
        for idx, rs in OS.backward_matching(stack, matchers, S):
            OS.update_trackers(rs, idx, trackers[name])
        bwd = np.stack([OS.rle_seg_to_pan_seg(rs, pans[0].shape) for rs in stack])
        OS.finish_tracking(trackers[name])
        per_axis[name] = dict(pan=np.stack(pans), fwd=fwd, bwd=bwd,
                              raw={t.class_id: {k: dict(v) for k, v in t.instances.items()} for t in trackers[name]})
        for t in trackers[name]:
            OS.remove_small_objects(t, min_size=100)
            OS.remove_pancakes(t, min_span=3)
    cons, vols = {}, {}
    for cid in labels:
        cts = [t for name in axes for t in trackers[name] if t.class_id == cid]
        if cid in thing:
            con = OC.create_instance_consensus(cts, 2, 0.75, False)
            OS.remove_small_objects(con, min_size=100)
            OS.remove_pancakes(con, min_span=3)
        else:
            con = OC.create_semantic_consensus(cts, 2)
        cons[cid] = con.instances
        vols[cid] = OR.numpy_fill_instances(np.zeros(shape, np.uint32), con.instances)
    return per_axis, cons, vols


def test_pipeline_matches_reference():
    g = load_golden('pipeline')
    for i in range(int(g['n'])):
        C, ks, _, head_seed = (int(x) for x in g[f'p{i}_par'])
        per_axis, cons, vols = run_oracle_pipeline(g[f'p{i}_lab'], g[f'p{i}_cls'], C, ks, head_seed)
        labels = [1] if C == 1 else list(range(1, C + 1))
        for name in ('xy', 'xz', 'yz'):
            np.testing.assert_array_equal(per_axis[name]['pan'], g[f'p{i}_{name}_pan'], err_msg=f'{i} {name} pan')
            np.testing.assert_array_equal(per_axis[name]['fwd'], g[f'p{i}_{name}_fwd'], err_msg=f'{i} {name} fwd')
            np.testing.assert_array_equal(per_axis[name]['bwd'], g[f'p{i}_{name}_bwd'], err_msg=f'{i} {name} bwd')
            for cid in labels:
                assert_instances_equal(per_axis[name]['raw'][cid], unpack_instances(g, f'p{i}_{name}_tr{cid}'))
        for cid in labels:
            assert_instances_equal(cons[cid], unpack_instances(g, f'p{i}_con{cid}'))
            np.testing.assert_array_equal(vols[cid], g[f'p{i}_vol{cid}'])
            if cid == 1:
                assert vols[cid].max() >= 3, "planted workload should produce several instances"


def test_pipeline_driver_serial_and_parallel_match_reference():
    """oracle/pipeline.py (the driver the full-size GPU tests and bench.py's cpu_baseline use), serial and with the
    per-pixel stages spread over worker processes, against the reference's own consensus volumes and panoptic slices."""
    from oracle import pipeline as PL
    g = load_golden('pipeline')
    for i in range(int(g['n'])):
        C, ks, _, head_seed = (int(x) for x in g[f'p{i}_par'])
        lab, cls = g[f'p{i}_lab'], g[f'p{i}_cls']
        thing = [1] if C == 1 else list(range(1, C))
        labels = [1] if C == 1 else list(range(1, C + 1))
        engine = dict(thing_list=thing, label_divisor=1000, stuff_area=16, void_label=0, nms_threshold=0.1, nms_kernel=7,
                      confidence_thr=0.5, median_kernel_size=ks)
        heads = {a: {k: v.numpy() for k, v in SY.planted_heads(lab, cls, a, n_classes=C, seed=head_seed,
                                                                coarse=False).items()} for a in PL.AXES}
        for workers in (1, 3):
            vols, n_inst, _ = PL.orthoplane_volume(heads, lab.shape, engine, dict(merge_iou_thr=0.25, merge_ioa_thr=0.25),
                                                   dict(min_size=100, min_span=3),
                                                   dict(pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False),
                                                   labels=labels, workers=workers)
            for cid in labels:
                np.testing.assert_array_equal(vols[cid], g[f'p{i}_vol{cid}'], err_msg=f'{i} class {cid} workers {workers}')
        h = heads['xz']
        pans, _ = PL.plane_pans(h['sem'], h['ctr_hmp'], h['offsets'], engine, labels=labels, workers=3)
        np.testing.assert_array_equal(np.stack(pans), g[f'p{i}_xz_pan'])
    # a stack shorter than the median kernel loses its tail in both forms (engines.py:68-90)
    short = {k: v[:4] for k, v in heads['xy'].items()}
    e7 = dict(engine, median_kernel_size=7)
    a, _ = PL.plane_pans(short['sem'], short['ctr_hmp'], short['offsets'], e7, labels=labels, workers=1)
    b, _ = PL.plane_pans(short['sem'], short['ctr_hmp'], short['offsets'], e7, labels=labels, workers=2)
    assert len(a) == len(b) == len(PL.emitted_slices(4, 7)) < 4
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
