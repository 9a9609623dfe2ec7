"""1x1 convolution + BN + residual + ReLU on the ResNet-50 conv3 shapes of the bench (32 slices of 1024^2 per call):
ms, TF/s, algorithmic GB/s.  Variants are selected by the experiment switches of emp_conv.hip (EMP_CONV_NO_WS: the
tiled kernel instead of the weight-stationary one of emp_conv1x1.hip on the layer1 / layer2 shapes; EMP_CONV_NARROW,
EMP_CONV_NO_RESPF, EMP_CONV_BK), one process per variant, all on the SAME device (A/B).
`python tools/bench_res1x1.py [quick]`"""
import os
import subprocess
import sys

CASES = [('l1.conv3 64->256 @256', 64, 256, 256, True), ('l1.shortcut 64->256 @256 (no res)', 64, 256, 256, False),
         ('l2.conv3 128->512 @128', 128, 512, 128, True),
         ('l3.conv3 256->1024 @64', 256, 1024, 64, True), ('l4.conv3 512->2048 @64', 512, 2048, 64, True)]
VARIANTS = [('shipped', {}), ('tiled kernel (EMP_CONV_NO_WS=1)', {'EMP_CONV_NO_WS': '1'}), ('shipped, again', {}),
            ('narrow', {'EMP_CONV_NO_WS': '1', 'EMP_CONV_NARROW': '1'}), ('no_respf', {'EMP_CONV_NO_WS': '1', 'EMP_CONV_NO_RESPF': '1'}),
            ('no_respf_bk16', {'EMP_CONV_NO_RESPF': '1', 'EMP_CONV_BK': '16'}),
            ('narrow_no_respf_bk16', {'EMP_CONV_NARROW': '1', 'EMP_CONV_NO_RESPF': '1', 'EMP_CONV_BK': '16'})]


def child():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from empanada_amd import _hip
    B = 32
    for name, cin, cout, hw, use_res in CASES:
        x = torch.randn(B, cin, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 1, 1, device='cuda') * 0.05).permute(0, 2, 3, 1).contiguous()
        sc, sh = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda')
        r = torch.randn(B, cout, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last) if use_res else None
        for _ in range(3):
            _hip.conv_bn_act_nhwc(x, w, sc, sh, r, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _hip.conv_bn_act_nhwc(x, w, sc, sh, r, True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        px = B * hw * hw
        print(f'  {name:26s} {ms:7.3f} ms {2 * px * cin * cout / ms / 1e9:6.1f} TF/s '
              f'{4 * px * (cin + (2 if use_res else 1) * cout) / ms / 1e6:6.0f} GB/s', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child()
    else:
        for name, env in (VARIANTS[:3] if 'quick' in sys.argv else VARIANTS):
            print(name, env, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env={**os.environ, **env}, check=True)
