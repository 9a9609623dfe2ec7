import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
args = bench.parse()
dev = torch.device('cuda', 0)
pipe = bench.Pipeline(args, dev)
x = torch.randn((8, 1, 1024, 1024), device=dev).contiguous(memory_format=torch.channels_last)
def run():
    with torch.no_grad():
        return {k: v.clone() for k, v in pipe.model(x).items()}
def cmp(tag):
    a = run(); b = run(); c = run()
    print(tag, {k: (bool(torch.equal(a[k], b[k])), bool(torch.equal(a[k], c[k])), float((a[k]-b[k]).abs().max())) for k in a}, flush=True)
cmp('miopen-only')
pipe.tune(1024)
cmp('tuned')
# per-site: find non-deterministic modules
mods = [(n, m) for n, m in pipe.model.named_modules() if isinstance(m, FusedConvBNAct)]
outs = {}
def hook(name):
    def f(m, i, o):
        t = o[0] if isinstance(o, (tuple, list)) else o
        outs.setdefault(name, []).append(t.detach().clone() if torch.is_tensor(t) else None)
    return f
hs = [m.register_forward_hook(hook(n)) for n, m in mods]
with torch.no_grad():
    pipe.model(x); pipe.model(x)
for n, m in mods:
    a, b = outs[n][0], outs[n][1]
    if a is not None and not torch.equal(a, b):
        print('NONDET', n, m.impl, tuple(a.shape), float((a-b).abs().max()), flush=True)
        break
