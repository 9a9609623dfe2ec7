import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden
from empanada_amd.models import PanopticBiFPN, prepare_for_inference, synthesize_weights
g = load_golden('models')
def build():
    m = synthesize_weights(PanopticBiFPN(encoder='regnety_6p4gf', num_classes=1)).eval()
    with torch.no_grad():
        for head in (m.semantic_head, m.ins_center, m.ins_xy): head.head[1].weight.mul_(1e-3)
    return m
x = torch.from_numpy(g['x'])
for fuse in (False, True):
    for cl in (True, False):
        m = prepare_for_inference(build(), 'cuda', fuse=fuse, channels_last=cl)
        xx = x.cuda().contiguous(memory_format=torch.channels_last) if cl else x.cuda()
        with torch.no_grad(): out = m(xx)
        print('fuse', fuse, 'channels_last', cl, {k: float(np.abs(out[k].float().cpu().numpy()-g[f'bifpn_regnety_{k}']).max()) for k in out})
# where does it diverge: encoder features
m = build(); mg = prepare_for_inference(build(), 'cuda', fuse=False)
with torch.no_grad():
    fc = m.encoder(x); fg = mg.encoder(x.cuda().contiguous(memory_format=torch.channels_last))
for i,(a,b) in enumerate(zip(fc,fg)):
    print('enc', i, float((a-b.cpu()).abs().max()), float(a.abs().max()))
