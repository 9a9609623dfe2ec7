"""A few launches of the MFMA kernel on representative layers for rocprofv3 --pmc passes.
usage: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_conv.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from empanada_amd import _hip

B = 32
CASES = [('head.pw 256->256 @128', 256, 256, 128, 1, 0, 1), ('l4.down 1024->2048 @32', 1024, 2048, 32, 1, 0, 1),
         ('aspp d2 2048->256 @32', 2048, 256, 32, 3, 2, 2), ('l1.conv1 256->64 @128', 256, 64, 128, 1, 0, 1),
         ('l2.conv1 512->128 @64', 512, 128, 64, 1, 0, 1)]
for name, cin, cout, hw, k, pad, dil in CASES:
    x = torch.randn(B, cin, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device='cuda') * 0.02).permute(0, 2, 3, 1).contiguous()
    sc, sh = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda')
    for _ in range(3):
        _hip.conv_bn_act_nhwc(x, w, sc, sh, None, True, 1, pad, dil)
    torch.cuda.synchronize()
    print(name, flush=True)
# batched GEMM of layer4's F(4,3): 36 x [2048 x 512] x [512 x 512]
V = torch.randn(36, 2048, 512, device='cuda')
U = torch.randn(36, 512, 512, device='cuda')
M = torch.empty(36, 2048, 512, device='cuda')
for _ in range(3):
    _hip.call('emp_gemm_nt_batched', V.data_ptr(), U.data_ptr(), 36, 2048, 512, 512, M.data_ptr(), _hip.stream())
torch.cuda.synchronize()
