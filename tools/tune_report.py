"""Print the per-call-site tuning decisions of the PanopticDeepLab forward (batch 32, 512^2).
usage: PYTHONPATH=. python tools/tune_report.py"""
import torch

from empanada_amd.models import PanopticDeepLab, prepare_for_inference, synthesize_weights, tune_fused_convs

torch.backends.cudnn.benchmark = True
model = prepare_for_inference(synthesize_weights(PanopticDeepLab(encoder='resnet50', num_classes=1)), 'cuda')
x = torch.rand(32, 1, 512, 512, device='cuda').contiguous(memory_format=torch.channels_last)
rep = tune_fused_convs(model, x, reps=10, verbose=True)
tot = {k: 0.0 for k in ('miopen', 'best')}
for name, (best, t) in rep.items():
    tot['miopen'] += t['miopen']
    tot['best'] += t[best]
print(tot)
