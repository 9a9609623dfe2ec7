"""emp_dwconv_nhwc timing on the decoder / head shapes.  usage: PYTHONPATH=. python tools/bench_dw.py"""
import torch
from empanada_amd import _hip
for (C, H) in ((256, 128), (288, 128), (320, 64), (384, 32)):
    x = torch.randn(32, C, H, H, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(25, C, device='cuda')
    for _ in range(3):
        _hip.dwconv_nhwc(x, w, None, 5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _hip.dwconv_nhwc(x, w, None, 5)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(C, H, round(ms, 4), 'ms', round(8 * x.numel() / ms / 1e6, 1), 'GB/s')
