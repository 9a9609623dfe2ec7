import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from empanada_amd.inference import sharded
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device('cuda', 0)
stacks, heads, n_obj, _ = bench.build_inputs_ortho(S, dev)
h = heads['xy']
pan = sharded.sharded_panoptic_stack(h['sem'], h['ctr_hmp'], h['offsets'], coarse_boundaries=False, **bench.ENGINE)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    table, host = sharded.sharded_tables(pan, [1], [1], bench.ENGINE['label_divisor'])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    pt = sharded.finish_plane(table, host, pan.shape[0], 'xy', (S, S, S), [1], [1], bench.ENGINE['label_divisor'], **bench.MATCH)
    torch.cuda.synchronize()
    pr.disable(); t2 = time.perf_counter()
    print(f'tables {t1-t0:.3f}s finish_plane {t2-t1:.3f}s n_comp {table.n_comp} n_runs {table.n_runs} inst {pt.n_inst}')
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
import numpy as np, ctypes
from empanada_amd.inference import patterns as PA
from empanada_amd import _hip
lib = _hip.load()
orig = lib.emp_chain_class
class W:
    t = 0.0
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig(*a); W.t += time.perf_counter() - t0; return r
lib_emp = W()
import types
_hip._lib.emp_chain_class = lib_emp
t0 = time.perf_counter()
PA.chain_from_tables(host, pan.shape[0], [1], [1], bench.ENGINE['label_divisor'], 0.25, 0.25)
print('chain total', time.perf_counter() - t0, 'native incl callbacks', W.t)
np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'gpurun_out', 'chain_tables.npz'), D=pan.shape[0], **host)
