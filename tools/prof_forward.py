"""Per-layer timing of the PanopticDeepLab forward on the GPU (hooks + events), batch 32 x 512 x 512.
usage: python tools/prof_forward.py [batch] [size]"""
import sys
import time

import torch

from empanada_amd.models import PanopticDeepLab, prepare_for_inference, synthesize_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.backends.cudnn.benchmark = True
model = PanopticDeepLab(encoder='resnet50', num_classes=1)
model = synthesize_weights(model)
model = prepare_for_inference(model, 'cuda', torch.float32)
x = torch.rand(B, 1, S, S, device='cuda').contiguous(memory_format=torch.channels_last)
recs = []


def pre(m, inp):
    e = torch.cuda.Event(enable_timing=True); e.record(); m._e0 = e


def post(name):
    def f(m, inp, out):
        e = torch.cuda.Event(enable_timing=True); e.record()
        fl = 0
        desc = type(m).__name__
        if isinstance(m, torch.nn.Conv2d):
            o = out
            fl = 2 * o.numel() * m.in_channels // m.groups * m.kernel_size[0] * m.kernel_size[1]
            desc = f"conv {m.in_channels}->{m.out_channels} k{m.kernel_size[0]} s{m.stride[0]} d{m.dilation[0]} g{m.groups} out{tuple(o.shape[2:])}"
        recs.append((name, desc, m._e0, e, fl))
    return f


for n, m in model.named_modules():
    if len(list(m.children())) == 0:
        m.register_forward_pre_hook(pre)
        m.register_forward_hook(post(n))
with torch.no_grad():
    for _ in range(2):
        recs.clear()
        torch.cuda.synchronize(); t0 = time.time()
        out = model(x)
        torch.cuda.synchronize(); t1 = time.time()
print(f"forward {1e3 * (t1 - t0):.1f} ms (with hooks)")
rows = [(n, d, e0.elapsed_time(e1), fl) for n, d, e0, e1, fl in recs]
tot = sum(r[2] for r in rows)
print(f"sum of leaf modules {tot:.1f} ms")
for n, d, t, fl in sorted(rows, key=lambda r: -r[2])[:45]:
    print(f"{t:8.3f} ms  {fl / t / 1e9 if fl else 0:7.1f} TF/s  {n:55s} {d}")
by = {}
for n, d, t, fl in rows:
    k = d.split(' ')[0] if d.startswith('conv') else d
    by[k] = by.get(k, 0) + t
print({k: round(v, 1) for k, v in sorted(by.items(), key=lambda kv: -kv[1])})
