"""Grouped 3x3 convolution (emp_gconv3x3_bn_act_nhwc) on the RegNetY-6.4GF stage shapes of a 512^2 tile, N slices
per call, against MIOpen's grouped convolution + the BN/ReLU epilogue pass.  `python tools/bench_gconv.py [N]`"""
import sys

import torch

from empanada_amd import _hip


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    print(f'{"shape":28s} {"G":>3s} {"s":>2s} {"hip ms":>8s} {"TF/s":>7s} {"miopen+bn ms":>13s} {"TF/s":>7s}')
    for (C, G, H, stride) in [(144, 2, 256, 2), (144, 2, 128, 1), (288, 4, 128, 2), (288, 4, 64, 1),
                              (576, 8, 64, 2), (576, 8, 32, 1), (1296, 18, 32, 2), (1296, 18, 16, 1)]:
        GW = C // G
        x = torch.randn(N, C, H, H, device='cuda').contiguous(memory_format=torch.channels_last)
        conv = torch.nn.Conv2d(C, C, 3, stride, 1, groups=G, bias=False).cuda().to(memory_format=torch.channels_last)
        w = conv.weight.detach().permute(0, 2, 3, 1).contiguous()
        sc, sh = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
        OH = (H - 1) // stride + 1
        flops = 2 * N * OH * OH * C * GW * 9
        with torch.no_grad():
            t_hip = timeit(lambda: _hip.gconv3x3_bn_act_nhwc(x, w, G, sc, sh, True, stride))
            t_lib = timeit(lambda: _hip.bn_act_nhwc_(conv(x), sc, sh, None, True, None))
            a = _hip.gconv3x3_bn_act_nhwc(x, w, G, sc, sh, True, stride)
            b = _hip.bn_act_nhwc_(conv(x), sc, sh, None, True, None)
            err = float((a - b).abs().max())
        print(f'{str((N, C, H, H)):28s} {G:3d} {stride:2d} {t_hip:8.3f} {flops / t_hip / 1e9:7.1f} {t_lib:13.3f} '
              f'{flops / t_lib / 1e9:7.1f}  max|diff| {err:.2e}', flush=True)


if __name__ == '__main__':
    main()
