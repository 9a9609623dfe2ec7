"""The fused ResNet stem (emp_stem_conv7_bn_relu_maxpool) against the two-step path it replaces (MIOpen convolution +
emp_bn_relu_maxpool_nhwc) at the bench's call shape (32 slices of 1024^2, or [N S]).  `PYTHONPATH=. python tools/bench_stem.py`"""
import sys

import torch

from empanada_amd import _hip

torch.backends.cudnn.benchmark = True


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
    N, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 1024)
    x = torch.randn(N, 1, S, S, device='cuda')
    conv = torch.nn.Conv2d(1, 64, 7, 2, 3, bias=False).cuda().to(memory_format=torch.channels_last)
    sc, sh = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda')
    w_tc = conv.weight.detach()[:, 0].reshape(64, 49).t().contiguous()
    xcl = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        t_conv = timeit(lambda: conv(xcl))
        t_two = timeit(lambda: _hip.bn_relu_maxpool_nhwc(conv(xcl), sc, sh))
        t_one = timeit(lambda: _hip.stem_conv7_bn_relu_maxpool(x, w_tc, sc, sh))
        a = _hip.bn_relu_maxpool_nhwc(conv(xcl), sc, sh)
        b = _hip.stem_conv7_bn_relu_maxpool(x, w_tc, sc, sh)
    flops = 2 * N * (S // 2) ** 2 * 64 * 49
    print(f'({N}, 1, {S}, {S}): MIOpen conv {t_conv:.3f} ms, conv + bn_relu_maxpool {t_two:.3f} ms, fused stem {t_one:.3f} ms '
          f'({flops / t_one / 1e9:.1f} TF/s on the vector ALUs), max|diff| {float((a - b).abs().max()):.2e}')


if __name__ == '__main__':
    main()
