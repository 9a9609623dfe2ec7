"""HIP-event timing of emp_upsample_bilinear on the decoder's shapes (32 slices of 1024^2 per call: 256-channel maps at
1/16 -> 1/8 -> 1/4 resolution, written into a channel slice of the concat buffer, and the ASPP image pooling 1 x 1 -> 64^2).
usage: python tools/bench_upsample.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from empanada_amd import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda')
cases = [(256, 64, 128, 320), (256, 128, 256, 288), (256, 1, 64, 1280)]     # C, in, out, channels of the concat buffer
for C, h, H, Ctot in cases:
    x = torch.randn(B, C, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    buf = torch.empty(B, Ctot, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    out = buf[:, :C]
    ref = torch.nn.functional.interpolate(x, size=(H, H), mode='bilinear', align_corners=True)
    _hip.upsample_bilinear(x, (H, H), out=out)
    err = (out - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _hip.upsample_bilinear(x, (H, H), out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = 4 * (x.numel() + out.numel()) / 1e9
    print(f'C={C} {h}^2 -> {H}^2 x{B}: {ms:.3f} ms, {gb / ms * 1e3:.0f} GB/s of algorithmic bytes, max |diff| vs torch {err:.2e}')
