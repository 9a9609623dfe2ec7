import time, sys, torch, numpy as np
sys.path.insert(0, '.')
from empanada_amd import synthetic as SY
t=time.perf_counter()
lab, cls = SY.planted_labels((32,512,512), fill=0.08, rmin=6, rmax=24, seed=4321)
print('labels', time.perf_counter()-t, len(cls)); t=time.perf_counter()
torch.cuda.init(); x=torch.zeros(1,device='cuda'); torch.cuda.synchronize()
print('cuda init', time.perf_counter()-t); t=time.perf_counter()
h = SY.planted_heads(lab, cls, 'xy', device='cuda', seed=99)
torch.cuda.synchronize()
print('heads', time.perf_counter()-t); t=time.perf_counter()
h = SY.planted_heads(lab, cls, 'xy', device='cuda', seed=99)
torch.cuda.synchronize()
print('heads again', time.perf_counter()-t); t=time.perf_counter()
