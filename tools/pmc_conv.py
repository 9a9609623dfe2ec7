"""A few launches of the MFMA kernels for rocprofv3 --pmc passes.
usage: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_conv.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from empanada_amd import _hip

B = 32
x = torch.randn(B, 2048, 32, 32, device='cuda').contiguous(memory_format=torch.channels_last)
w = (torch.randn(256, 2048, 3, 3, device='cuda') * 0.02)
w_okkc = w.permute(0, 2, 3, 1).contiguous()
sc, sh = torch.rand(256, device='cuda') + 0.5, torch.randn(256, device='cuda')
for _ in range(3):
    _hip.conv_bn_act_nhwc(x, w_okkc, sc, sh, None, True, 1, 2, 2)
tiles = torch.from_numpy(_hip.wino_tiles(B, 32, 32, 2)).cuda()
U = _hip.wino_filter_transform(w).cuda()
for _ in range(3):
    _hip.wino_conv_bn_act(x, U, tiles, 2, sc, sh, True)
x1 = torch.randn(B, 64, 128, 128, device='cuda').contiguous(memory_format=torch.channels_last)
w1 = (torch.randn(256, 64, 1, 1, device='cuda') * 0.1).permute(0, 2, 3, 1).contiguous()
r1 = torch.randn(B, 256, 128, 128, device='cuda').contiguous(memory_format=torch.channels_last)
for _ in range(3):
    _hip.conv_bn_act_nhwc(x1, w1, sc, sh, r1, True, 1, 0, 1)
torch.cuda.synchronize()
