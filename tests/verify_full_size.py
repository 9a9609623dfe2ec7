"""Full-size parity run (BASELINE configs[3], the metric's volume): the whole 1024^3 orthoplane consensus volume of the
HIP path against the CPU oracle (oracle/pipeline.py, per-pixel stages on the host cores), instance ids included.
Too long for the default test suite (about five minutes of CPU work, ~80 GB of host memory): run by hand, its verdict
goes to profiles/ --
  python tests/verify_full_size.py [S] > profiles/r3_verify_ortho<S>.json
-- or as a test: EMP_VERIFY_FULL=1 python -m pytest tests/test_full_size_gpu.py -m gpu -k whole_1024"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def log(msg):
    print(f'[{time.perf_counter() - T0:7.1f}s] {msg}', file=sys.stderr, flush=True)


T0 = time.perf_counter()


def run(S=1024):
    from empanada_amd import _hip
    from oracle import consensus as OC
    from oracle import pipeline as PL
    from oracle import rle_ops as OR
    from oracle import rle_seg as OS
    _hip.load()
    dev = torch.device('cuda')
    shape = (S, S, S)
    stacks, heads, n_obj, _ = bench.build_inputs_ortho(S, dev)
    log(f'inputs ready: {n_obj} planted objects')
    t0 = time.perf_counter()
    n_found, vols, _ = bench.postprocess_planes(heads, shape, None, {})
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    got = vols[1].view(torch.int32).cpu().numpy().view(np.uint32)
    del vols
    log(f'HIP path: {n_found} consensus instances in {t_gpu:.2f} s')
    workers = min(16, os.cpu_count() or 1)
    trackers, timers = {}, {}
    t_cpu0 = time.perf_counter()
    for axis in PL.AXES:
        h = {k: v.cpu().numpy() for k, v in heads[axis].items()}
        del heads[axis]
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        _, rles = PL.plane_pans(h['sem'], h['ctr_hmp'], h['offsets'], bench.ENGINE, labels=[1], workers=workers)
        t1 = time.perf_counter()
        del h
        log(f'oracle {axis}: pixels + rle {t1 - t0:.1f} s')
        trackers[axis] = PL.plane_trackers(rles, axis, shape, [1], [1], bench.ENGINE['label_divisor'], bench.MATCH,
                                           bench.FILTERS)
        del rles
        timers[axis] = [round(t1 - t0, 1), round(time.perf_counter() - t1, 1)]
        log(f'oracle {axis}: matching + trackers {timers[axis][1]} s, {sum(len(t.instances) for t in trackers[axis])} instances')
    t0 = time.perf_counter()
    con = OC.create_instance_consensus([t for a in PL.AXES for t in trackers[a]], bench.CONSENSUS['pixel_vote_thr'],
                                       bench.CONSENSUS['cluster_iou_thr'], bench.CONSENSUS['bypass'])
    log(f'oracle consensus {time.perf_counter() - t0:.1f} s')
    OS.remove_small_objects(con, bench.FILTERS['min_size'])
    OS.remove_pancakes(con, bench.FILTERS['min_span'])
    exp = OR.numpy_fill_instances(np.zeros(shape, np.uint32), con.instances)
    t_cpu = time.perf_counter() - t_cpu0
    same = bool(np.array_equal(got, exp))
    n_diff = int((got != exp).sum()) if not same else 0
    res = {'volume': list(shape), 'planted_objects': n_obj, 'hip_consensus_instances': int(n_found),
           'oracle_consensus_instances': int(len(con.instances)), 'volumes_identical_ids_included': same,
           'differing_voxels': n_diff, 'labelled_voxels': int((exp > 0).sum()),
           'hip_postprocessing_s': round(t_gpu, 2), 'oracle_postprocessing_s': round(t_cpu, 1), 'oracle_workers': workers,
           'oracle_per_plane_s(pixels+rle, match+track)': timers,
           'what': 'planted heads of bench.py (seeds 1234 / 4321 / 99 + s), MitoNet engine parameters; HIP: '
                   'bench.postprocess_planes; oracle: oracle/pipeline.py plane by plane + oracle consensus + fill'}
    log('identical' if same else f'DIFFERENT in {n_diff} voxels')
    return res


if __name__ == '__main__':
    result = run(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
    print(json.dumps(result), flush=True)
    sys.exit(0 if result['volumes_identical_ids_included'] else 1)
