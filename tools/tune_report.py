"""Print the per-call-site tuning decisions of the PanopticDeepLab forward (batch 32, 512^2) with the
direct-convolution-equivalent TFLOP/s of every candidate and each site's share of the tuned forward.
usage: PYTHONPATH=. python tools/tune_report.py [batch size [pdl_r50 | bifpn_r50 | bifpn_regnety]]"""
import sys

import torch

from empanada_amd.models import PanopticBiFPN, PanopticDeepLab, prepare_for_inference, synthesize_weights, tune_fused_convs
from empanada_amd.models.panoptic_deeplab import FusedConvBNAct

torch.backends.cudnn.benchmark = True
which = sys.argv[3] if len(sys.argv) > 3 else 'pdl_r50'
plain = {'pdl_r50': lambda: PanopticDeepLab(encoder='resnet50', num_classes=1),
         'bifpn_r50': lambda: PanopticBiFPN(encoder='resnet50', num_classes=1),
         'bifpn_regnety': lambda: PanopticBiFPN(encoder='regnety_6p4gf', num_classes=1)}[which]()
model = prepare_for_inference(synthesize_weights(plain), 'cuda')
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 512)
x = torch.rand(B, 1, S, S, device="cuda").contiguous(memory_format=torch.channels_last)
rep = tune_fused_convs(model, x, reps=10)
mods = dict(model.named_modules())
rows = []
for name, (best, t) in rep.items():
    m = mods[name]
    assert isinstance(m, FusedConvBNAct)
    (n, cin, h, w), has_res = m._seen
    conv = m.conv
    co, _, kh, kw = conv.weight.shape
    sh, sw = conv.stride
    ho, wo = (h + sh - 1) // sh, (w + sw - 1) // sw
    flops = 2.0 * n * ho * wo * co * cin * kh * kw / conv.groups
    byts = 4.0 * (n * h * w * cin + n * ho * wo * co * (2 if has_res else 1))
    rows.append((name, (n, cin, h, w), co, kh, sh, conv.dilation[0], has_res, best, t, flops, byts))
total = sum(r[8][r[7]] for r in rows)
print(f'{"site":48s} {"in":>18s} {"co":>5s} k s d res {"best":>8s} {"ms":>7s} {"%":>5s} {"TF/s":>6s} {"GB/s":>6s}  others')
for name, shp, co, k, s, d, res, best, t, flops, byts in rows:
    others = ' '.join(f'{i}={v:.3f}' for i, v in t.items() if i != best)
    print(f'{name:48s} {str(shp[1:]):>18s} {co:5d} {k} {s} {d} {int(res)}   {best:>8s} {t[best]:7.3f} '
          f'{100 * t[best] / total:5.1f} {flops / t[best] / 1e9:6.1f} {byts / t[best] / 1e6:6.0f}  {others}')
print(f'total tuned {total:.2f} ms, MIOpen only {sum(r[8]["miopen"] for r in rows):.2f} ms, '
      f'{sum(r[9] for r in rows) / total / 1e9:.1f} TF/s overall')
