"""emp_conv_bn_act_nhwc vs MIOpen conv2d + emp_bn_act_nhwc on the PanopticDeepLab/ResNet-50 layer shapes (batch 32, 512^2).
usage: PYTHONPATH=. python tools/bench_conv.py"""
import torch
import torch.nn.functional as F

from empanada_amd import _hip

torch.backends.cudnn.benchmark = True
B = 32
# name, H(in), Cin, Cout, k, stride, pad, dil, residual
SHAPES = [
    ('l1.conv1 1x1 256->64', 128, 256, 64, 1, 1, 0, 1, False),
    ('l1.conv2 3x3 64->64', 128, 64, 64, 3, 1, 1, 1, False),
    ('l1.conv3 1x1 64->256 +res', 128, 64, 256, 1, 1, 0, 1, True),
    ('l2.conv1 1x1 512->128', 64, 512, 128, 1, 1, 0, 1, False),
    ('l2.conv2 3x3 128->128', 64, 128, 128, 3, 1, 1, 1, False),
    ('l2.conv3 1x1 128->512 +res', 64, 128, 512, 1, 1, 0, 1, True),
    ('l2.0.conv2 3x3 s2 128->128', 128, 128, 128, 3, 2, 1, 1, False),
    ('l2.0.down 1x1 s2 256->512', 128, 256, 512, 1, 2, 0, 1, False),
    ('l3.conv1 1x1 1024->256', 32, 1024, 256, 1, 1, 0, 1, False),
    ('l3.conv2 3x3 256->256', 32, 256, 256, 3, 1, 1, 1, False),
    ('l3.conv3 1x1 256->1024 +res', 32, 256, 1024, 1, 1, 0, 1, True),
    ('l4.conv1 1x1 2048->512', 32, 2048, 512, 1, 1, 0, 1, False),
    ('l4.conv2 3x3 d2 512->512', 32, 512, 512, 3, 1, 2, 2, False),
    ('l4.conv3 1x1 512->2048 +res', 32, 512, 2048, 1, 1, 0, 1, True),
    ('l4.0.down 1x1 1024->2048', 32, 1024, 2048, 1, 1, 0, 1, False),
    ('aspp 3x3 d6 2048->256', 32, 2048, 256, 3, 1, 6, 6, False),
    ('aspp 1x1 2048->256', 32, 2048, 256, 1, 1, 0, 1, False),
    ('aspp.project 1x1 1280->256', 32, 1280, 256, 1, 1, 0, 1, False),
    ('fuse2.pw 1x1 288->256', 128, 288, 256, 1, 1, 0, 1, False),
    ('head.pw 1x1 256->256', 128, 256, 256, 1, 1, 0, 1, False),
    ('proj 1x1 256->32', 128, 256, 32, 1, 1, 0, 1, False),
]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"{'layer':34s} {'miopen+bn ms':>12s} {'(conv only)':>11s} {'emp fused ms':>12s} {'TF/s':>7s} {'speedup':>8s}")
tot_a = tot_b = 0
for name, H, Cin, Cout, k, s, p, d, res in SHAPES:
    x = torch.randn(B, Cin, H, H, device='cuda').contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device='cuda') * 0.05).contiguous(memory_format=torch.channels_last)
    w_okkc = w.permute(0, 2, 3, 1).contiguous()
    sc, sh = torch.rand(Cout, device='cuda') + 0.5, torch.randn(Cout, device='cuda')
    y0 = F.conv2d(x, w, None, s, p, d)
    r = torch.randn_like(y0) if res else None

    def ref():
        y = F.conv2d(x, w, None, s, p, d)
        if Cout % 4 == 0:
            return _hip.bn_act_nhwc_(y, sc, sh, r, True)
        return torch.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))

    def conv_only():
        return F.conv2d(x, w, None, s, p, d)

    def mine():
        return _hip.conv_bn_act_nhwc(x, w_okkc, sc, sh, r, True, s, p, d)

    wino_ms = None
    if k == 3 and s == 1 and p == d:
        tiles = torch.from_numpy(_hip.wino_tiles(B, H, H, d)).cuda()
        U = _hip.wino_filter_transform(w).cuda()

        def wino():
            return _hip.wino_conv_bn_act(x, U, tiles, d, sc, sh, True)

        wino_ms = timeit(wino)
        tiles4 = torch.from_numpy(_hip.wino_tiles(B, H, H, d, 4)).cuda()
        U4 = _hip.wino4_filter_transform(w).cuda()

        def wino4():
            return _hip.wino4_conv_bn_act(x, U4, tiles4, d, sc, sh, True)

        w4_ms = timeit(wino4)
        tiles3 = torch.from_numpy(_hip.wino_tiles(B, H, H, d, 3)).cuda()
        U3 = _hip.wino3_filter_transform(w).cuda()
        w3_ms = timeit(lambda: _hip.wino3_conv_bn_act(x, U3, tiles3, d, sc, sh, True))
        w3err = (ref() - _hip.wino3_conv_bn_act(x, U3, tiles3, d, sc, sh, True)).abs().max().item()
        w4err = (ref() - wino4()).abs().max().item()
        _hip.PROFILE = {}
        wino()
        torch.cuda.synchronize()
        parts = {kk: round(v[0][0].elapsed_time(v[0][1]), 3) for kk, v in _hip.PROFILE.items()}
        _hip.PROFILE = None
        werr = (ref() - wino()).abs().max().item()
    a, c, b = timeit(ref), timeit(conv_only), timeit(mine)
    err = (ref() - mine()).abs().max().item()
    fl = 2.0 * y0.numel() * Cin * k * k
    tot_a += a
    tot_b += min(a, b)
    print(f"{name:34s} {a:12.3f} {c:11.3f} {b:12.3f} {fl / b / 1e9:7.1f} {a / b:8.2f}   maxerr {err:.2e}")
    if wino_ms is not None:
        print(f"{'   winograd':34s} {wino_ms:12.3f}  speedup vs miopen+bn {a / wino_ms:.2f}  parts {parts}  maxerr {werr:.2e}")
        _hip.PROFILE = {}
        wino4()
        torch.cuda.synchronize()
        parts4 = {kk: round(v[0][0].elapsed_time(v[0][1]), 3) for kk, v in _hip.PROFILE.items()}
        _hip.PROFILE = None
        print(f"{'   winograd F(4,3)':34s} {w4_ms:12.3f}  speedup vs miopen+bn {a / w4_ms:.2f}  parts {parts4}  maxerr {w4err:.2e}")
        print(f"{'   winograd F(3,3)':34s} {w3_ms:12.3f}  speedup vs miopen+bn {a / w3_ms:.2f}  maxerr {w3err:.2e}")
        tot_b += min(a, b, wino_ms, w4_ms, w3_ms) - min(a, b)
print(f"sum miopen+bn {tot_a:.2f} ms; sum best-of {tot_b:.2f} ms")
