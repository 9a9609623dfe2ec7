"""CPU: the dense-path oracles (oracle/dense.py, plain C / numpy restatements with a fixed summation order) against the
library calls the reference makes for the same operations (torch conv2d / interpolate / max_pool2d on the host), within
the fp32 tolerances the GPU tests state.  The GPU tests then require the HIP kernels to equal these oracles bit for bit."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dense as OD


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def _wino_tiles(N, H, W, dil, m):
    from empanada_amd._hip import wino_tiles          # pure numpy helper (no library call)
    return wino_tiles(N, H, W, dil, m)


@pytest.mark.parametrize('slab', [16, 32])
@pytest.mark.parametrize('cfg', [(2, 9, 11, 64, 24, 1, 1, 0, 1), (1, 12, 10, 32, 8, 3, 1, 2, 2), (1, 11, 9, 32, 8, 3, 2, 1, 1)])
def test_conv_oracle_vs_torch(cfg, slab):
    N, H, W, Cin, Cout, k, stride, pad, dil = cfg
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, None, stride, pad, dil)
    res = torch.randn(ref.shape, generator=g)
    got = OD.conv_bn_act_nhwc(_nhwc(x), _nhwc(w), sc.numpy(), sh.numpy(), _nhwc(res), True, stride, pad, dil, slab)
    exp = torch.relu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    bound = F.conv2d(x.abs(), w.abs(), None, stride, pad, dil) * sc.view(1, -1, 1, 1)
    assert np.all(np.abs(got - _nhwc(exp)) <= 2e-6 * _nhwc(bound) + 1e-6)


@pytest.mark.parametrize('cfg', [(1, 9, 7, 64, 24, 3, 1, 1, 1, 5), (2, 6, 6, 96, 20, 1, 1, 0, 1, 3),
                                 (1, 11, 8, 32, 12, 3, 2, 2, 2, 4)])
def test_conv_splitk_oracle_vs_torch(cfg):
    """D4c: partial chains per K range added in ascending order -- within fp32 rounding of torch's conv2d, and with ONE
    range exactly the plain oracle with a K-slab of 32"""
    N, H, W, Cin, Cout, k, stride, pad, dil, ks = cfg
    g = torch.Generator().manual_seed(Cin + Cout + k + ks)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    ref = F.conv2d(x, w, None, stride, pad, dil)
    res = torch.randn(ref.shape, generator=g)
    got = OD.conv_splitk_bn_act_nhwc(_nhwc(x), _nhwc(w), sc.numpy(), sh.numpy(), _nhwc(res), True, stride, pad, dil, ks)
    exp = torch.relu(ref * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    bound = F.conv2d(x.abs(), w.abs(), None, stride, pad, dil) * sc.view(1, -1, 1, 1)
    assert np.all(np.abs(got - _nhwc(exp)) <= 2e-6 * _nhwc(bound) + 1e-6)
    one = OD.conv_splitk_bn_act_nhwc(_nhwc(x), _nhwc(w), sc.numpy(), sh.numpy(), _nhwc(res), True, stride, pad, dil, 1)
    plain = OD.conv_bn_act_nhwc(_nhwc(x), _nhwc(w), sc.numpy(), sh.numpy(), _nhwc(res), True, stride, pad, dil, 32)
    np.testing.assert_array_equal(one.view(np.uint32), plain.view(np.uint32))


@pytest.mark.parametrize('m,tol', [(2, 1e-5), (3, 2e-5), (4, 2e-5)])
@pytest.mark.parametrize('dil', [1, 2, 6])
def test_winograd_oracles_vs_torch(m, tol, dil):
    N, H, W, Cin, Cout = 1, 13, 9, 32, 8
    g = torch.Generator().manual_seed(m * 10 + dil)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    tiles = _wino_tiles(N, H, W, dil, m)
    fn = {2: OD.wino_conv_bn_act, 3: OD.wino3_conv_bn_act, 4: OD.wino4_conv_bn_act}[m]
    got = fn(_nhwc(x), w.numpy(), tiles, dil)
    ref = F.conv2d(x, w, None, padding=dil, dilation=dil)
    bound = F.conv2d(x.abs(), w.abs(), None, padding=dil, dilation=dil)
    assert np.all(np.abs(got - _nhwc(ref)) <= tol * _nhwc(bound) + 1e-6)


def test_dwconv_upsample_pointwise_maxpool_oracles_vs_torch():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 9, 7, generator=g)
    w = torch.randn(8, 1, 5, 5, generator=g) * 0.2
    got = OD.dwconv_nhwc(_nhwc(x), w.reshape(8, 25).t().contiguous().numpy())
    assert np.abs(got - _nhwc(F.conv2d(x, w, None, padding=2, groups=8))).max() < 1e-5
    up = OD.upsample_bilinear(x.numpy(), (18, 21))
    assert np.abs(up - F.interpolate(x, size=(18, 21), mode='bilinear', align_corners=True).numpy()).max() < 1e-5
    pw, pb = torch.randn(2, 8, generator=g), torch.randn(2, generator=g)
    po = OD.pointwise_out_nhwc(_nhwc(x), pw.numpy(), pb.numpy())
    assert np.abs(po - F.conv2d(x, pw.view(2, 8, 1, 1), pb).numpy()).max() < 1e-5
    sc, sh = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g)
    mp = OD.bn_relu_maxpool_nhwc(_nhwc(x), sc.numpy(), sh.numpy())
    ref = F.max_pool2d(torch.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1)
    np.testing.assert_array_equal(mp, _nhwc(ref))


def test_projection_oracle_vs_torch():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 32, 6, 5, generator=g)
    w = torch.randn(128, 32, 1, 1, generator=g) * 0.2
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    pw, pb = torch.randn(2, 128, generator=g) * 0.1, torch.randn(2, generator=g)
    got = OD.conv_bn_act_proj_nhwc(_nhwc(x), _nhwc(w), sc.numpy(), sh.numpy(), True, pw.numpy(), pb.numpy())
    y = torch.relu(F.conv2d(x, w) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref = F.conv2d(y, pw.view(2, 128, 1, 1), pb).numpy()
    assert np.abs(got - ref).max() < 1e-4


@pytest.mark.parametrize('cfg', [(2, 9, 11, 2, 72, 1), (1, 8, 8, 4, 72, 2), (2, 7, 5, 3, 56, 1), (1, 9, 9, 2, 112, 2),
                                 (1, 6, 6, 2, 16, 1)])
def test_gconv_oracle_vs_torch(cfg):
    """oracle.dense.gconv3x3_bn_act_nhwc (the order emp_gconv3x3_bn_act_nhwc documents) against the library call the
    reference makes -- Conv2d(groups=G) + BatchNorm2d(eval) + ReLU (encoders/regnet.py:59-71) -- in float64:
    |err| <= 2e-6 * sum|x||w| * |scale| + 1e-6."""
    N, H, W, G, GW, stride = cfg
    C = G * GW
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, GW, 3, 3, generator=g) * (1.0 / (GW * 9) ** 0.5)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    got = torch.from_numpy(OD.gconv3x3_bn_act_nhwc(_nhwc(x), w.permute(0, 2, 3, 1).contiguous().numpy(),
                                                   G, sc.numpy(), sh.numpy(), True, stride)).permute(0, 3, 1, 2)
    ref = F.conv2d(x.double(), w.double(), None, stride=stride, padding=1, groups=G)
    ref = torch.relu(ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    bound = F.conv2d(x.abs(), w.abs(), None, stride=stride, padding=1, groups=G) * sc.view(1, -1, 1, 1)
    assert torch.all((got.double() - ref).abs() <= 2e-6 * bound + 1e-6)


def test_gate_oracle_vs_torch():
    """oracle conv with the squeeze-excite gate epilogue against x * sigmoid(conv1x1(s) + b) (blocks.py:35-50)."""
    g = torch.Generator().manual_seed(5)
    s_in = torch.relu(torch.randn(2, 48, 6, 7, generator=g))
    w = torch.randn(144, 48, 1, 1, generator=g) * 0.2
    b = torch.randn(144, generator=g)
    x = torch.randn(2, 144, 6, 7, generator=g)
    got = torch.from_numpy(OD.conv_bn_act_nhwc(_nhwc(s_in), w.permute(0, 2, 3, 1).contiguous().numpy(),
                                               None, b.numpy(), _nhwc(x), 'gate', slab=16)).permute(0, 3, 1, 2)
    ref = x.double() * torch.sigmoid(F.conv2d(s_in.double(), w.double(), b.double()))
    assert torch.all((got.double() - ref).abs() <= 2e-6 * x.abs() + 1e-7)


@pytest.mark.parametrize('shape', [(2, 37, 45), (1, 64, 64), (1, 9, 7)])
def test_stem_oracle_vs_torch(shape):
    """oracle.dense.stem_conv7_bn_relu_maxpool against conv1 -> bn1 -> relu -> maxpool of the reference's ResNet
    (encoders/resnet.py:186-188,217-222) evaluated by torch in float64: |err| <= 2e-6 * sum|x||w| * |scale| + 1e-6."""
    N, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, 1, H, W, generator=g)
    w = torch.randn(64, 1, 7, 7, generator=g) * (1.0 / 7)
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.5
    got = torch.from_numpy(OD.stem_conv7_bn_relu_maxpool(x[:, 0].numpy(), w[:, 0].reshape(64, 49).t().contiguous().numpy(),
                                                         sc.numpy(), sh.numpy())).permute(0, 3, 1, 2)
    conv = F.conv2d(x.double(), w.double(), stride=2, padding=3)
    ref = F.max_pool2d(torch.relu(conv * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)), 3, 2, 1)
    bound = F.max_pool2d(F.conv2d(x.abs(), w.abs(), stride=2, padding=3) * sc.view(1, -1, 1, 1), 3, 2, 1)
    assert got.shape == ref.shape and torch.all((got.double() - ref).abs() <= 2e-6 * bound + 1e-6)
