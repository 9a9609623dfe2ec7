"""Reference-vs-port CPU timing (BUILD CONTAINER ONLY: imports the reference from /root/reference with the stand-ins of
oracle/gen_golden.py; never runs on the GPU box).  `python -m oracle.time_reference [side ...]` from the repo root.

What BASELINE.md section 3, step 1 promised: on identical inputs, the REFERENCE's own call sequence
(scripts/pdl_inference3d.py:140-233 without the process plumbing: PanopticDeepLabRenderEngine3d -> pan_seg_to_rle_seg ->
apply_matchers -> backward_matching -> trackers -> filters -> instance consensus -> filters -> fill) next to this
repository's CPU restatement of it (oracle/pipeline.py, one worker), which is what bench.py's `cpu_baseline` times on
the GPU box -- so that the port's Mvox/s can be read as reference Mvox/s through the recorded ratio.  Both legs get the
bench's recipe (planted ellipsoids, heads derived as in empanada_amd/synthetic.py, MitoNet engine parameters) on an
n^3 orthoplane volume, post-processing only (the conv forward is the same torch-CPU library call on both sides and is
timed separately on one 512 x 512 tile = BASELINE configs[0]).  The two result volumes must be identical.

Caveat printed with the numbers: the reference's numba kernels run here as plain Python (numba is absent; the stand-in
is the identity decorator), so the reference leg is SLOWER than a real installation wherever numba matters
(array_utils.rle_voting, chunk_ranges, box_iou); skimage's label / regionprops are stood in by scipy.ndimage."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import _install_standins      # noqa: E402

ENGINE = dict(thing_list=[1], label_divisor=20000, stuff_area=64, void_label=0, nms_threshold=0.1, nms_kernel=7,
              confidence_thr=0.3, median_kernel_size=7)
MATCH = dict(merge_iou_thr=0.25, merge_ioa_thr=0.25)
FILTERS = dict(min_size=500, min_span=4)
CONSENSUS = dict(pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False)


def reference_volume(heads, shape):
    import torch
    from empanada import array_utils as AU
    from empanada.inference import engines as EN
    from empanada.inference import filters as FI
    from empanada.inference import patterns as PA
    from empanada.inference import rle as RL

    class Stub(torch.nn.Module):
        def __init__(self, h):
            super().__init__()
            self.h, self.t = h, 0
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x, render_steps=2, interpolate_ins=True):
            t, self.t = self.t, self.t + 1
            return {'sem_logits': self.h['sem'][t:t + 1], 'ctr_hmp': self.h['ctr_hmp'][t:t + 1],
                    'offsets': self.h['offsets'][t:t + 1]}

    orig = EN.logits_to_prob
    EN.logits_to_prob = lambda x: x                   # the planted `sem` already is a probability
    div, thing = ENGINE['label_divisor'], ENGINE['thing_list']
    trackers = PA.create_axis_trackers({'xy': 0, 'xz': 1, 'yz': 2}, [1], div, shape)
    t_eng = t_rest = 0.0
    for axis in ('xy', 'xz', 'yz'):
        h = heads[axis]
        S, _, H, W = h['sem'].shape
        t0 = time.perf_counter()
        eng = EN.PanopticDeepLabRenderEngine3d(Stub(h), padding_factor=1, coarse_boundaries=False, **ENGINE)
        pans = []
        for t in range(S):
            o = eng(torch.zeros(1, 1, H, W), (H, W))
            if o is not None:
                pans.append(o.squeeze().numpy())
        pans += [o.squeeze().numpy() for o in eng.end()]
        t1 = time.perf_counter()
        matchers = PA.create_matchers(thing, div, MATCH['merge_iou_thr'], MATCH['merge_ioa_thr'])
        stack = [PA.apply_matchers(RL.pan_seg_to_rle_seg(p, [1], div, thing, force_connected=True), matchers)
                 for p in pans]
        for idx, rs in PA.backward_matching(stack, matchers, S):
            PA.update_trackers(rs, idx, trackers[axis])
        PA.finish_tracking(trackers[axis])
        for tr in trackers[axis]:
            FI.remove_small_objects(tr, min_size=FILTERS['min_size'])
            FI.remove_pancakes(tr, min_span=FILTERS['min_span'])
        t_eng += t1 - t0
        t_rest += time.perf_counter() - t1
    t0 = time.perf_counter()
    con = PA.create_instance_consensus(PA.get_axis_trackers_by_class(trackers, 1), CONSENSUS['pixel_vote_thr'],
                                       CONSENSUS['cluster_iou_thr'], CONSENSUS['bypass'])
    FI.remove_small_objects(con, min_size=FILTERS['min_size'])
    FI.remove_pancakes(con, min_span=FILTERS['min_span'])
    vol = np.zeros(shape, dtype=np.uint32)
    AU.numpy_fill_instances(vol, con.instances)
    EN.logits_to_prob = orig
    return vol, {'engine_s': t_eng, 'rle_match_track_s': t_rest, 'consensus_fill_s': time.perf_counter() - t0}


def main():
    _install_standins()
    import torch
    from empanada_amd import synthetic as SY
    from oracle import pipeline as PL
    sides = [int(a) for a in sys.argv[1:]] or [64, 128]
    torch.set_num_threads(os.cpu_count() or 1)
    out = {'host': f'{os.cpu_count()} cores (build container)', 'orthoplane': []}
    for n in sides:
        shape = (n, n, n)
        lab, cls = SY.planted_labels(shape, fill=0.08, rmin=6, rmax=24, seed=4321)
        heads = {a: SY.planted_heads(lab, cls, a, seed=99) for a in ('xy', 'xz', 'yz')}
        t0 = time.perf_counter()
        ref_vol, parts = reference_volume(heads, shape)
        t_ref = time.perf_counter() - t0
        np_heads = {a: {k: v.numpy() for k, v in heads[a].items()} for a in heads}
        t0 = time.perf_counter()
        tm = {}
        vols, n_inst, _ = PL.orthoplane_volume(np_heads, shape, ENGINE, MATCH, FILTERS, CONSENSUS, labels=[1], workers=1,
                                               timers=tm)
        t_port = time.perf_counter() - t0
        same = bool(np.array_equal(ref_vol, vols[1]))
        out['orthoplane'].append({'side': n, 'objects': int(n_inst), 'volumes_identical': same,
                                  'reference_s': round(t_ref, 2), 'port_s': round(t_port, 2),
                                  'reference_over_port': round(t_ref / t_port, 2),
                                  'reference_Mvox_s': round(n ** 3 / t_ref / 1e6, 4),
                                  'port_Mvox_s': round(n ** 3 / t_port / 1e6, 4),
                                  'reference_parts_s': {k: round(v, 2) for k, v in parts.items()},
                                  'port_parts_s': {k: round(v, 2) for k, v in tm.items()}})
        print(json.dumps(out['orthoplane'][-1]), flush=True)
        assert same, 'the port and the reference disagree'
    # cfg 1: one 512 x 512 tile through the MitoNet model + Render engine, reference classes vs this package's (CPU)
    from empanada.inference.engines import PanopticDeepLabRenderEngine as RefEngine
    from empanada.models.quantization.panoptic_deeplab import QuantizablePanopticDeepLabPR as RefQPR
    from empanada_amd.models import PanopticDeepLabPR, synthesize_weights
    from oracle.gen_golden_r4 import DAMP, MITO
    from oracle.gen_golden_r4 import ENGINE as ENG1
    ours = synthesize_weights(PanopticDeepLabPR(**MITO)).eval()
    with torch.no_grad():
        for head, damp in DAMP.items():
            getattr(ours, head).head[1].weight.mul_(damp)
    ref = RefQPR(quantize=False, **MITO)
    ref.load_state_dict(ours.state_dict(), strict=True)
    ref.eval()
    rng = np.random.default_rng(512)
    img = np.clip(rng.normal(129.8, 37.9, (512, 512)), 0, 255).astype(np.uint8)
    x = ((torch.from_numpy(img).float() - 255 * 0.508979) / (255 * 0.148561))[None, None]
    res = {}
    for name, model in (('reference', ref), ('port', ours)):
        with torch.no_grad():
            model(x, 2, False)
            t0 = time.perf_counter()
            for _ in range(3):
                o = model(x, 2, False)
            res[f'{name}_forward_s'] = round((time.perf_counter() - t0) / 3, 3)
    eng = RefEngine(ref, **ENG1)
    with torch.no_grad():
        t0 = time.perf_counter()
        pan = eng(x, (512, 512))
        res['reference_engine_call_s'] = round(time.perf_counter() - t0, 3)
    from oracle import postprocess as OP
    sem = OP.logits_to_prob(o['sem_logits'].numpy())
    t0 = time.perf_counter()
    kw = {k: v for k, v in ENG1.items() if k != 'padding_factor'}
    opan = OP.post_slice({'sem': sem, 'ctr_hmp': o['ctr_hmp'].numpy(), 'offsets': o['offsets'].numpy(),
                          'size': (512, 512)}, render=True, **kw)
    res['port_postprocess_s'] = round(time.perf_counter() - t0, 3)
    res['pan_identical'] = bool(np.array_equal(np.asarray(opan).squeeze(), pan.numpy().squeeze()))
    res['objects'] = int(len(np.unique(pan.numpy())) - 1)
    out['cfg1_tile_512'] = res
    print(json.dumps(res), flush=True)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles',
                        'r3_reference_vs_port_cpu.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print('written', path)


if __name__ == '__main__':
    main()
