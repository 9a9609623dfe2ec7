import sys, time, torch
sys.path.insert(0, '.')
from bench import build_inputs, ENGINE
from empanada_amd import _hip
from empanada_amd.inference.postprocess import centers_batched
dev = torch.device('cuda', 0)
vol, heads, n = build_inputs(256, 512, dev)
sem = _hip.median_harden_stack(heads['sem'], 7, 0.3)
idx, cnt = centers_batched(heads['ctr_hmp'], 0.1, 7)
print('K mean', float(cnt.float().mean()), 'max', int(cnt.max()), 'thing frac', float(sem.float().mean()))
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
off = heads['offsets']
print('normal            us', t(lambda: _hip.group_pixels(idx, cnt, off, 1, sem=sem, thing_list=[1])))
print('K = 0             us', t(lambda: _hip.group_pixels(idx, torch.zeros_like(cnt), off, 1, sem=sem, thing_list=[1])))
print('K = 1             us', t(lambda: _hip.group_pixels(idx, torch.ones_like(cnt), off, 1, sem=sem, thing_list=[1])))
print('no thing pixels   us', t(lambda: _hip.group_pixels(idx, cnt, off, 1, sem=torch.zeros_like(sem), thing_list=[1])))
print('all pixels voted  us', t(lambda: _hip.group_pixels(idx, cnt, off, 1, sem=None), 3))
