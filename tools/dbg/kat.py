import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from conftest import load_golden, unpack_instances
from empanada_amd import consensus as CO
from empanada_amd.inference import rle, tracker
g = load_golden('consensus_kat')
vols = g['vols']
shape = vols[0].shape
trs = [tracker.InstanceTracker(1, 1000, shape, axis='xy') for _ in range(3)]
for v, tr in zip(vols, trs):
    segs, _ = rle.stack_to_rle_segs(torch.from_numpy(v.astype(np.int32)).cuda().view(torch.uint32), [1], 1000, [1], force_connected=False)
    for z in range(shape[0]):
        tr.update(segs[z][1], z)
    tr.finish()
for t in trs:
    print('tracker', {k: (v['box'], int(v['runs'].sum())) for k, v in t.instances.items()})
for j in range(6):
    vote, iou_thr, bypass = g[f'k{j}_par']
    print('case', j, vote, iou_thr, bypass)
    try:
        inst = CO.merge_objects_from_trackers(trs, int(vote), float(iou_thr), bool(bypass))
    except Exception as e:
        print('  EXC', repr(e)); continue
    exp = unpack_instances(g, f'k{j}_inst')
    print('  got', {k: (v['box'], int(v['runs'].sum()), len(v['runs'])) for k, v in inst.items()})
    print('  exp', {k: (tuple(int(x) for x in v['box']), int(v['runs'].sum()), len(v['runs'])) for k, v in exp.items()})
