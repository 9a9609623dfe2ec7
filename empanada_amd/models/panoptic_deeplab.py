"""Panoptic-DeepLab with a ResNet encoder.

State-dict compatible re-implementation of the reference model family (cited per class):
  encoder   ResNet, 1-channel stem, dilated layer4 for output stride 16   empanada/models/encoders/resnet.py:143-232
  ASPP                                                                    empanada/models/decoders/aspp.py:22-102
  decoder   project low-level features, bilinear(align_corners) up, 5x5 separable fuse
                                                                          empanada/models/decoders/panoptic_deeplab.py:23-80
  heads     5x5 separable conv + BN + ReLU -> 1x1 conv                    empanada/models/heads.py:9-19
  model     PanopticDeepLab.forward -> {'sem_logits','ctr_hmp','offsets'} empanada/models/panoptic_deeplab.py:20-115
The module attribute names reproduce the reference's parameter names (e.g.
``semantic_decoder.fuse.0.0.sepconv.1.weight``); nothing else is shared with it.
"""
import zlib
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ['PanopticDeepLab', 'PanopticDeepLabPR', 'PointRendSemSegHead', 'resnet_encoder', 'prepare_for_inference',
           'tune_fused_convs',
           'synthesize_weights']

_RESNETS = {
    'resnet18': ('basic', [2, 2, 2, 2]), 'resnet34': ('basic', [3, 4, 6, 3]),
    'resnet50': ('bottleneck', [3, 4, 6, 3]), 'resnet101': ('bottleneck', [3, 4, 23, 3]),
    'resnet152': ('bottleneck', [3, 8, 36, 3]),
}


def _conv_bn_act(nin, nout, k, stride=1, groups=1, act=True):
    layers = [nn.Conv2d(nin, nout, k, stride=stride, padding=(k - 1) // 2, groups=groups, bias=False),
              nn.BatchNorm2d(nout)]
    if act:
        layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class SeparableConv2d(nn.Module):
    """depthwise k x k + pointwise 1 x 1 (blocks.py:15-35); parameters live under `.sepconv.{0,1}`"""

    def __init__(self, nin, nout, kernel_size=3, stride=1, bias=True):
        super().__init__()
        self.sepconv = nn.Sequential(
            nn.Conv2d(nin, nin, kernel_size, stride=stride, padding=(kernel_size - 1) // 2, groups=nin, bias=bias),
            nn.Conv2d(nin, nout, 1, stride=1, bias=bias))

    def forward(self, x):
        return self.sepconv(x)


def _sepconv_bn_act(nin, nout, k):
    return nn.Sequential(SeparableConv2d(nin, nout, k, 1, bias=False), nn.BatchNorm2d(nout), nn.ReLU(inplace=True))


def _bn_affine(bn):
    """eval-mode BatchNorm as a per-channel affine map: (scale, shift) fp32"""
    scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)).contiguous()
    shift = (bn.bias.detach().float() - bn.running_mean.detach().float() * scale).contiguous()
    return scale, shift


def _no_late_weights(state_dict, prefix, *args):
    """The inference stand-ins fold their weights at construction (BatchNorm -> scale / shift, filters permuted or
    Winograd-transformed): loading a state dict into a prepared model would leave those stale.  Fail loudly."""
    raise RuntimeError("load_state_dict after prepare_for_inference: load the weights into the plain model first "
                       f"(module {prefix.rstrip('.')!r} holds folded copies)")


class DepthwiseConvNHWC(nn.Module):
    """Inference-only stand-in for Conv2d(C, C, k, padding=k//2, groups=C) on NHWC fp32 activations:
    emp_dwconv_nhwc (one HBM pass) instead of MIOpen's grouped-convolution kernel."""

    def __init__(self, conv):
        super().__init__()
        k = conv.kernel_size[0]
        C = conv.in_channels
        # (C, 1, k, k) -> (k*k, C)
        self.register_buffer('w_kkc', conv.weight.detach().float().reshape(C, k * k).t().contiguous())
        self.register_buffer('b', conv.bias.detach().float().contiguous() if conv.bias is not None else None)
        self.k = k

    @staticmethod
    def eligible(m):
        return (isinstance(m, nn.Conv2d) and m.groups == m.in_channels == m.out_channels and m.in_channels % 4 == 0
                and m.kernel_size[0] == m.kernel_size[1] and m.kernel_size[0] in (3, 5) and m.stride == (1, 1)
                and m.dilation == (1, 1) and m.padding == (m.kernel_size[0] // 2,) * 2
                and m.padding_mode == 'zeros')

    def forward(self, x):
        from .. import _hip
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        return _hip.dwconv_nhwc(x, self.w_kkc, self.b, self.k)


class FusedBNAct(nn.Module):
    """Inference-only stand-in for BatchNorm2d(+ReLU): one in-place HIP pass (emp_bn_act_nhwc) over the NHWC
    conv output instead of separate MIOpen batch-norm and ReLU kernels."""

    def __init__(self, bn, relu):
        super().__init__()
        scale, shift = _bn_affine(bn)
        self.register_buffer('scale', scale, persistent=False)
        self.register_buffer('shift', shift, persistent=False)
        self.relu = relu
        self._register_load_state_dict_pre_hook(_no_late_weights)

    def forward(self, x, residual=None, out=None):
        from .. import _hip
        if not (x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0):
            raise RuntimeError("FusedBNAct needs fp32 CUDA activations with C % 4 == 0")
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
            residual = residual.contiguous(memory_format=torch.channels_last)
        return _hip.bn_act_nhwc_(x, self.scale, self.shift, residual, self.relu, out)


class PointwiseOutNHWC(nn.Module):
    """Inference-only stand-in for the heads' last Conv2d(C, n <= 4, 1, bias): emp_pointwise_out_nhwc, a streaming
    pass instead of a GEMM-shaped MIOpen kernel.  Output is planar (N, n, H, W)."""

    def __init__(self, conv):
        super().__init__()
        self.register_buffer('w', conv.weight.detach().float().reshape(conv.out_channels, conv.in_channels).contiguous())
        self.register_buffer('b', conv.bias.detach().float().contiguous() if conv.bias is not None else None)

    @staticmethod
    def eligible(m):
        return (isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0)
                and m.groups == 1 and m.out_channels <= 4 and m.in_channels % 4 == 0 and m.in_channels <= 1024)

    def forward(self, x):
        from .. import _hip
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        return _hip.pointwise_out_nhwc(x, self.w, self.b)


class FusedConvBNAct(nn.Module):
    """Conv2d + FusedBNAct (+ residual) as one call with three interchangeable implementations:
      'miopen' : MIOpen convolution, then the emp_bn_act_nhwc epilogue pass              (default)
      'direct' : emp_conv_bn_act_nhwc -- implicit GEMM on the fp32 matrix cores, epilogue fused
      'direct_sk': emp_conv_splitk_bn_act_nhwc -- the same kernel with the reduction cut into ranges, one block per
                 range, for launches that would leave most CUs without a block (one 512^2 tile at batch 1: layer3 /
                 layer4 / ASPP); a candidate only where emp_conv_splitk_plan says so
      'wino'   : Winograd F(2x2,3x3), input transform inside the GEMM loader -- emp_wino_gemm_fused /
                 emp_wino_output_transform
      'wino4'  : Winograd F(4x4,3x3), 36 GEMMs, 4x fewer matrix-core FLOPs; rounding error about 10x the direct
                 form's (emp_wino4_input_transform / emp_gemm_nt_batched / emp_wino4_output_transform)
      'wino3'  : Winograd F(3x3,3x3), 25 GEMMs (for sub-grids of 5-6 rows: the dilation-6 ASPP branch)
      'wino_sep': the same with V materialised -- emp_wino_input_transform / emp_gemm_nt_batched /
                 emp_wino_output_transform (less L2 traffic per matrix-core FLOP; wins when Cin is large)
      'grouped': emp_gconv3x3_bn_act_nhwc -- the grouped 3x3 convolution of the RegNet bottleneck (one group per block,
                 16x16x4 matrix-core tiles cut to the group width)
    tune_fused_convs() times the candidates on the layer's real shape and keeps the fastest."""

    def __init__(self, conv, bn):
        super().__init__()
        self.conv, self.bn = conv, bn
        self.impl = 'miopen'
        self.tuned = {}
        self._seen = None
        self._w_okkc = None
        self._U = None
        self._U4 = None
        self._U3 = None
        self._tiles = {}

    def candidates(self, has_residual, shape=None):
        """shape: the (N, Cin, H, W) the site is called with -- lets the small-launch form in when it applies"""
        c = self.conv
        out = ['miopen']
        square = (c.bias is None and c.padding_mode == 'zeros' and c.kernel_size[0] == c.kernel_size[1]
                  and c.stride[0] == c.stride[1] and c.padding[0] == c.padding[1] and c.dilation[0] == c.dilation[1]
                  and c.weight.dtype == torch.float32)
        if square and c.groups == 1 and c.in_channels % 16 == 0:
            out.append('direct')
            if shape is not None and c.in_channels % 32 == 0 and c.out_channels % 4 == 0 and self.bn.relu in (0, 1, False, True):
                from .. import _hip
                k, st, pd, dl = c.kernel_size[0], c.stride[0], c.padding[0], c.dilation[0]
                oh = (shape[2] + 2 * pd - dl * (k - 1) - 1) // st + 1
                ow = (shape[3] + 2 * pd - dl * (k - 1) - 1) // st + 1
                if _hip.conv_splitk_plan(shape[0] * oh * ow, c.out_channels, c.in_channels, k, k) > 1:
                    out.append('direct_sk')
            if (c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == c.dilation and c.out_channels % 4 == 0
                    and c.in_channels % 32 == 0 and not has_residual):
                out.extend(['wino', 'wino_sep', 'wino4', 'wino3'])
        gw = c.in_channels // c.groups
        if (square and c.groups > 1 and c.in_channels == c.out_channels and c.kernel_size == (3, 3) and c.padding == (1, 1)
                and c.dilation == (1, 1) and c.stride[0] in (1, 2) and gw % 8 == 0 and 8 <= gw <= 128 and not has_residual):
            out.append('grouped')
        return out

    def _prepare(self, impl):
        from .. import _hip
        if impl in ('direct', 'direct_sk', 'grouped') and self._w_okkc is None:
            self._w_okkc = self.conv.weight.detach().permute(0, 2, 3, 1).contiguous()
        if impl in ('wino', 'wino_sep') and self._U is None:
            self._U = _hip.wino_filter_transform(self.conv.weight.detach())
        if impl == 'wino4' and self._U4 is None:
            self._U4 = _hip.wino4_filter_transform(self.conv.weight.detach()).to(self.conv.weight.device)
        if impl == 'wino3' and self._U3 is None:
            self._U3 = _hip.wino3_filter_transform(self.conv.weight.detach()).to(self.conv.weight.device)

    def release(self, keep):
        if keep not in ('direct', 'direct_sk', 'grouped'):
            self._w_okkc = None
        if keep not in ('wino', 'wino_sep'):
            self._U = None
        if keep != 'wino4':
            self._U4 = None
        if keep != 'wino3':
            self._U3 = None
        if keep not in ('wino', 'wino_sep', 'wino4', 'wino3'):
            self._tiles = {}

    def forward(self, x, residual=None, out=None):
        self._seen = (tuple(x.shape), residual is not None)
        impl = self.impl
        if impl == 'miopen' or not (x.is_cuda and x.dtype == torch.float32):
            return self.bn(self.conv(x), residual, out)
        from .. import _hip
        c = self.conv
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        self._prepare(impl)
        if impl in ('direct', 'direct_sk'):
            if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
                residual = residual.contiguous(memory_format=torch.channels_last)
            if impl == 'direct_sk':
                oh = (x.shape[2] + 2 * c.padding[0] - c.dilation[0] * (c.kernel_size[0] - 1) - 1) // c.stride[0] + 1
                ow = (x.shape[3] + 2 * c.padding[0] - c.dilation[0] * (c.kernel_size[0] - 1) - 1) // c.stride[0] + 1
                ks = _hip.conv_splitk_plan(x.shape[0] * oh * ow, c.out_channels, c.in_channels, *c.kernel_size)
                if ks > 1:                 # a bigger batch than the one the site was tuned on fills the chip: plain form
                    return _hip.conv_splitk_bn_act_nhwc(x, self._w_okkc, self.bn.scale, self.bn.shift, residual,
                                                        self.bn.relu, c.stride[0], c.padding[0], c.dilation[0], out,
                                                        k_splits=ks)
            return _hip.conv_bn_act_nhwc(x, self._w_okkc, self.bn.scale, self.bn.shift, residual, self.bn.relu,
                                         c.stride[0], c.padding[0], c.dilation[0], out)
        assert residual is None
        if impl == 'grouped':
            return _hip.gconv3x3_bn_act_nhwc(x, self._w_okkc, c.groups, self.bn.scale, self.bn.shift, self.bn.relu,
                                             c.stride[0], out)
        m = {'wino4': 4, 'wino3': 3}.get(impl, 2)
        key = (x.shape[0], x.shape[2], x.shape[3], m)
        if key not in self._tiles:
            self._tiles[key] = torch.from_numpy(_hip.wino_tiles(key[0], key[1], key[2], c.dilation[0], m)).to(x.device)
        if impl == 'wino3':
            return _hip.wino3_conv_bn_act(x, self._U3, self._tiles[key], c.dilation[0], self.bn.scale, self.bn.shift,
                                          self.bn.relu, out)
        if impl == 'wino4':
            return _hip.wino4_conv_bn_act(x, self._U4, self._tiles[key], c.dilation[0], self.bn.scale, self.bn.shift,
                                          self.bn.relu, out)
        return _hip.wino_conv_bn_act(x, self._U, self._tiles[key], c.dilation[0], self.bn.scale, self.bn.shift,
                                     self.bn.relu, out, fused=(impl == 'wino'))


def _up_bilinear(x, size, hip_ops, out=None):
    """F.interpolate(mode='bilinear', align_corners=True), through emp_upsample_bilinear when hip_ops"""
    if hip_ops and x.is_cuda and x.dtype == torch.float32:
        from .. import _hip
        return _hip.upsample_bilinear(x, size, out=out)
    y = F.interpolate(x, size=size, mode='bilinear', align_corners=True)
    if out is not None:
        out.copy_(y)
        return out
    return y


def _conv_bn_into(seq, x, out):
    """seq = Sequential(Conv2d, FusedBNAct(+ReLU), Identity...): the fused epilogue writes straight into `out`,
    a channel slice of the caller's concat buffer"""
    if isinstance(seq[0], FusedConvBNAct):
        seq[0](x, out=out)
    else:
        seq[1](seq[0](x), out=out)
    for extra in list(seq)[2:]:
        assert isinstance(extra, (nn.Identity, nn.Dropout)), "unexpected module after the fused epilogue"


def _can_write_into(seq, x):
    if not (isinstance(seq, nn.Sequential) and len(seq) >= 2 and x.is_cuda and x.dtype == torch.float32):
        return False
    return isinstance(seq[0], FusedConvBNAct) or (isinstance(seq[0], nn.Conv2d) and isinstance(seq[1], FusedBNAct))


class _Basic(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        if isinstance(self.conv1, FusedConvBNAct):
            return self.conv2(self.conv1(x), idt)
        if isinstance(self.bn1, FusedBNAct):
            out = self.bn1(self.conv1(x))
            return self.bn2(self.conv2(out), idt)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class _Bottleneck(nn.Module):
    """ResNet v1.5 bottleneck: stride on the 3x3 (resnet.py:89-130)"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        if isinstance(self.conv1, FusedConvBNAct):
            return self.conv3(self.conv2(self.conv1(x)), idt)
        if isinstance(self.bn1, FusedBNAct):
            out = self.bn2(self.conv2(self.bn1(self.conv1(x))))
            return self.bn3(self.conv3(out), idt)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class _Cfg:
    def __init__(self):
        self.widths = []
        self.w_stem = 64


class ResNetEncoder(nn.Module):
    """returns [p1 (1/4, stem), p2 (1/4), p3 (1/8), p4 (1/16), p5 (1/16 dilated | 1/32)] (resnet.py:217-229)"""

    def __init__(self, kind, layers, in_channels=1, output_stride=32):
        super().__init__()
        assert output_stride in (16, 32)
        block = _Basic if kind == 'basic' else _Bottleneck
        self.cfg = _Cfg()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._layer(block, 64, layers[0])
        self.layer2 = self._layer(block, 128, layers[1], stride=2)
        self.layer3 = self._layer(block, 256, layers[2], stride=2)
        last_stride, dil = (1, 2) if output_stride == 16 else (2, 1)
        self.layer4 = self._layer(block, 512, layers[3], stride=last_stride, dilation=dil)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _layer(self, block, planes, blocks, stride=1, dilation=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, down, dilation)]
        self.inplanes = planes * block.expansion
        self.cfg.widths.append(self.inplanes)
        mods += [block(self.inplanes, planes, dilation=dilation) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    hip_ops = False
    _stem_w = None                    # (49, 64) filter table of the fused stem, built at the first call

    def forward(self, x):
        if (self.hip_ops and isinstance(self.bn1, FusedBNAct) and x.is_cuda and x.dtype == torch.float32
                and self.maxpool.kernel_size == 3 and self.maxpool.stride == 2 and self.maxpool.padding == 1):
            from .. import _hip
            c1 = self.conv1
            if (c1.in_channels == 1 and c1.out_channels == 64 and c1.kernel_size == (7, 7) and c1.stride == (2, 2)
                    and c1.padding == (3, 3) and c1.bias is None and c1.weight.dtype == torch.float32):
                # the whole stem in one kernel (emp_stem_conv7_bn_relu_maxpool): the 1/2-resolution activation never
                # leaves the CU.  (N,1,H,W) is the same memory in NCHW and NHWC.
                if self._stem_w is None:
                    self._stem_w = c1.weight.detach()[:, 0].reshape(64, 49).t().contiguous()
                xs = x if x.is_contiguous() else x.contiguous()
                p1 = _hip.stem_conv7_bn_relu_maxpool(xs, self._stem_w, self.bn1.scale, self.bn1.shift)
            else:
                y = self.conv1(x)
                if not y.is_contiguous(memory_format=torch.channels_last):
                    y = y.contiguous(memory_format=torch.channels_last)
                p1 = _hip.bn_relu_maxpool_nhwc(y, self.bn1.scale, self.bn1.shift)   # stem epilogue in one pass
        elif isinstance(self.bn1, FusedBNAct):
            p1 = self.maxpool(self.bn1(self.conv1(x)))
        else:
            p1 = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        p2 = self.layer1(p1)
        p3 = self.layer2(p2)
        p4 = self.layer3(p3)
        p5 = self.layer4(p4)
        return [p1, p2, p3, p4, p5]


def resnet_encoder(name, output_stride=32, in_channels=1):
    kind, layers = _RESNETS[name]
    return ResNetEncoder(kind, layers, in_channels, output_stride)


class _ASPPPooling(nn.Module):
    def __init__(self, nin, nout):
        super().__init__()
        self.aspp_pooling = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(nin, nout, 1, bias=False), nn.ReLU())
        self.hip_ops = False

    def forward(self, x, out=None):
        size = x.shape[-2:]
        if self.hip_ops and x.is_cuda and x.dtype == torch.float32:
            from .. import _hip
            # mean over the image, then the 1x1 convolution as one call of the fused conv kernel on an N x 1 x 1
            # "image": the library routes this N x Cin GEMM to a split-K kernel that accumulates with atomics, i.e.
            # in a different order from run to run -- the only source of run-to-run differences in the forward
            conv = self.aspp_pooling[1]
            if getattr(self, '_w_okkc', None) is None or self._w_okkc.device != x.device:
                self._w_okkc = conv.weight.detach().permute(0, 2, 3, 1).contiguous()
            pooled = x.mean(dim=(2, 3), keepdim=True).contiguous(memory_format=torch.channels_last)
            # N pixels x 2048 channels -> 256: four tiles whose blocks would each walk all 64 K-slabs (73 us); cut into
            # ranges it is one short round of blocks (deterministic too: partial sums added in a fixed order)
            ks = _hip.conv_splitk_plan(pooled.shape[0], conv.out_channels, conv.in_channels, 1, 1) \
                if conv.in_channels % 32 == 0 and conv.out_channels % 4 == 0 else 1
            if ks > 1:
                y = _hip.conv_splitk_bn_act_nhwc(pooled, self._w_okkc, relu=True, k_splits=ks)
            else:
                y = _hip.conv_bn_act_nhwc(pooled, self._w_okkc, relu=True)
            return _up_bilinear(y, size, True, out)
        return _up_bilinear(self.aspp_pooling(x), size, self.hip_ops, out)


class ASPP(nn.Module):
    """1x1 + three dilated 3x3 + image pooling -> concat -> 1x1 project (aspp.py:51-102)"""

    def __init__(self, nin, nout, atrous_rates=(2, 4, 6), dropout_p=0.5):
        super().__init__()
        mods = [nn.Sequential(nn.Conv2d(nin, nout, 1, bias=False), nn.BatchNorm2d(nout), nn.ReLU())]
        for r in atrous_rates:
            mods.append(nn.Sequential(nn.Conv2d(nin, nout, 3, padding=r, dilation=r, bias=False),
                                      nn.BatchNorm2d(nout), nn.ReLU()))
        mods.append(_ASPPPooling(nin, nout))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(5 * nout, nout, 1, bias=False), nn.BatchNorm2d(nout), nn.ReLU(),
                                     nn.Dropout(dropout_p))
        self.nout = nout
        self.hip_ops = False

    def forward(self, x):
        if self.hip_ops and all(_can_write_into(c, x) for c in list(self.convs)[:-1]):
            # the concat buffer is allocated once and every branch's epilogue writes its own channel slice
            n = self.nout
            buf = torch.empty((x.shape[0], len(self.convs) * n, x.shape[2], x.shape[3]), dtype=x.dtype,
                              device=x.device, memory_format=torch.channels_last)
            for i, conv in enumerate(self.convs):
                dst = buf[:, i * n:(i + 1) * n]
                if isinstance(conv, _ASPPPooling):
                    conv(x, out=dst)
                else:
                    _conv_bn_into(conv, x, dst)
            return self.project(buf)
        return self.project(torch.cat([conv(x) for conv in self.convs], dim=1))


class PanopticDeepLabDecoder(nn.Module):
    """decoders/panoptic_deeplab.py:23-80"""

    def __init__(self, in_channels, decoder_channels, low_level_stages, low_level_channels,
                 low_level_channels_project, atrous_rates, aspp_channels=None, aspp_dropout=0.5):
        super().__init__()
        aspp_channels = aspp_channels or decoder_channels
        self.aspp = ASPP(in_channels, aspp_channels, atrous_rates, aspp_dropout)
        self.low_level_stages = list(low_level_stages)
        project, fuse = [], []
        for i, (lc, pc) in enumerate(zip(low_level_channels, low_level_channels_project)):
            project.append(_conv_bn_act(lc, pc, 1))
            fuse.append(_sepconv_bn_act((aspp_channels if i == 0 else decoder_channels) + pc, decoder_channels, 5))
        self.project = nn.ModuleList(project)
        self.fuse = nn.ModuleList(fuse)
        self.hip_ops = False

    def forward(self, pyramid: List[torch.Tensor]):
        x = self.aspp(pyramid[-1])
        for stage, proj, fuse in zip(self.low_level_stages, self.project, self.fuse):
            feat = pyramid[stage]
            if self.hip_ops and _can_write_into(proj, feat):
                # up-sampled x and the projected low-level features land directly in the concat buffer
                cx = x.shape[1]
                cl = (proj[0].conv if isinstance(proj[0], FusedConvBNAct) else proj[0]).out_channels
                buf = torch.empty((x.shape[0], cx + cl, feat.shape[2], feat.shape[3]), dtype=x.dtype,
                                  device=x.device, memory_format=torch.channels_last)
                _up_bilinear(x, feat.shape[2:], True, out=buf[:, :cx])
                _conv_bn_into(proj, feat, buf[:, cx:])
                x = fuse(buf)
            else:
                low = proj(feat)
                x = _up_bilinear(x, low.shape[2:], self.hip_ops)
                x = fuse(torch.cat((x, low), dim=1))
        return x


class PanopticDeepLabHead(nn.Module):
    """heads.py:9-19"""

    def __init__(self, nin, n_classes):
        super().__init__()
        self.head = nn.Sequential(_sepconv_bn_act(nin, nin, 5), nn.Conv2d(nin, n_classes, 1, bias=True))
        self.hip_ops = False

    def forward(self, x):
        if self.hip_ops and x.is_cuda and x.dtype == torch.float32:
            # depthwise -> [pointwise conv + BN + ReLU + last 1x1 conv] in one launch: the 256-channel activation
            # between the last two layers is never written (emp_conv_bn_act_proj_nhwc)
            sep = self.head[0][0]
            site = sep.sepconv[1] if isinstance(sep, SeparableConv2d) else None
            last = self.head[1]
            if (isinstance(site, FusedConvBNAct) and site.impl == 'direct' and isinstance(last, PointwiseOutNHWC)
                    and site.conv.out_channels in (128, 256) and site.conv.kernel_size == (1, 1)):
                from .. import _hip
                y = sep.sepconv[0](x)
                if not y.is_contiguous(memory_format=torch.channels_last):
                    y = y.contiguous(memory_format=torch.channels_last)
                site._prepare('direct')
                return _hip.conv_bn_act_proj_nhwc(y, site._w_okkc, site.bn.scale, site.bn.shift, site.bn.relu,
                                                  last.w, last.b)
        return self.head(x)


class PanopticDeepLab(nn.Module):
    """models/panoptic_deeplab.py:20-115.  forward(x (N,1,H,W)) -> dict of full-resolution heads."""

    def __init__(self, encoder='resnet50', num_classes=1, stage4_stride=16, decoder_channels=256,
                 low_level_stages=(3, 2, 1), low_level_channels_project=(128, 64, 32), atrous_rates=(2, 4, 6),
                 aspp_channels=None, aspp_dropout=0.1, ins_decoder=False, ins_ratio=0.5, **kwargs):
        super().__init__()
        assert encoder in _RESNETS, f'Invalid encoder name {encoder}, choices are {sorted(_RESNETS)}'
        assert stage4_stride in (16, 32) and min(low_level_stages) > 0
        self.decoder_channels = decoder_channels
        self.num_classes = num_classes
        self.encoder = resnet_encoder(encoder, output_stride=stage4_stride)
        sem_p, ins_p = (aspp_dropout, aspp_dropout) if isinstance(aspp_dropout, float) else aspp_dropout
        widths = self.encoder.cfg.widths
        low_ch = [int(widths[i - 1]) for i in low_level_stages]
        self.semantic_decoder = PanopticDeepLabDecoder(int(widths[-1]), decoder_channels, low_level_stages, low_ch,
                                                       list(low_level_channels_project), atrous_rates,
                                                       aspp_channels, sem_p)
        if ins_decoder:
            self.instance_decoder = PanopticDeepLabDecoder(
                int(widths[-1]), decoder_channels, low_level_stages, low_ch,
                [int(s * ins_ratio) for s in low_level_channels_project], atrous_rates, aspp_channels, ins_p)
        else:
            self.instance_decoder = None
        self.semantic_head = PanopticDeepLabHead(decoder_channels, num_classes)
        self.ins_center = PanopticDeepLabHead(decoder_channels, 1)
        self.ins_xy = PanopticDeepLabHead(decoder_channels, 2)

    hip_ops = False

    def _up4(self, x):
        # scale_factor=4 with align_corners=True: out = 4 * in, src = dst * (in - 1) / (out - 1)
        return _up_bilinear(x, (4 * x.shape[2], 4 * x.shape[3]), self.hip_ops)

    def forward(self, x):
        pyramid = self.encoder(x)
        sem_x = self.semantic_decoder(pyramid)
        ins_x = sem_x if self.instance_decoder is None else self.instance_decoder(pyramid)
        return {'sem_logits': self._up4(self.semantic_head(sem_x)),
                'ctr_hmp': self._up4(self.ins_center(ins_x)),
                'offsets': self._up4(self.ins_xy(ins_x))}


# ----------------------------------------------------------------------------- PointRend (inference)
class StandardPointHead(nn.Module):
    """Point MLP of 1x1 Conv1d layers; the coarse prediction is re-fed to every layer (point_rend.py:138-190)."""

    def __init__(self, nin, num_classes, fc_dim, num_fc):
        super().__init__()
        dim_in = nin + num_classes
        layers = []
        for _ in range(num_fc):
            layers.append(nn.Sequential(nn.Conv1d(dim_in, fc_dim, 1), nn.ReLU(inplace=True)))
            dim_in = fc_dim + num_classes
        self.fc_layers = nn.ModuleList(layers)
        self.predictor = nn.Conv1d(dim_in, num_classes, 1)

    def forward(self, fine, coarse):
        x = torch.cat([fine, coarse], dim=1)
        for layer in self.fc_layers:
            x = torch.cat([layer(x), coarse], dim=1)
        return self.predictor(x)


def _point_sample(features, coords):
    """bilinear sampling at [0,1]^2 points (N,P,2 as x,y) -> (N,C,P)  (point_rend.py:35-60)"""
    return F.grid_sample(features, 2.0 * coords.unsqueeze(2) - 1.0, mode='bilinear', align_corners=False).squeeze(3)


class PointRendSemSegHead(nn.Module):
    """Eval-mode PointRend refinement (point_rend.py:241-269): `subdivision_steps` times upsample x2, pick the
    `subdivision_num_points` most uncertain grid points, re-predict them with the point head from the fine
    features + coarse logits and scatter them back.  (The training branch is out of scope.)"""

    def __init__(self, nin, num_classes, num_fc=3, train_num_points=1024, oversample_ratio=3,
                 importance_sample_ratio=0.75, subdivision_steps=2, subdivision_num_points=8192, **kwargs):
        super().__init__()
        self.subdivision_steps = subdivision_steps
        self.subdivision_num_points = subdivision_num_points
        self.point_head = StandardPointHead(nin, num_classes, nin, num_fc)

    @staticmethod
    def _uncertainty(logits):
        if logits.size(1) == 1:
            return -logits.abs()
        top2 = torch.topk(logits, k=2, dim=1)[0]
        return (top2[:, 1] - top2[:, 0]).unsqueeze(1)

    hip_ops = False                                  # set by prepare_for_inference

    def _hip_weights(self):
        """Conv1d (Cout, Cin, 1) weights of the point head as (Cout, 1, 1, ld) with Cin zero-padded to a multiple of 16
        (the K granule of emp_conv_bn_act_nhwc) + biases, built once"""
        cached = getattr(self, '_hipw', None)
        dev = self.point_head.predictor.weight.device
        if cached is not None and cached[0].device == dev:
            return cached
        convs = [layer[0] for layer in self.point_head.fc_layers] + [self.point_head.predictor]
        cin = convs[0].weight.shape[1]
        ld = (cin + 15) // 16 * 16
        out = []
        for conv in convs:
            w = torch.zeros((conv.weight.shape[0], 1, 1, ld), dtype=torch.float32, device=dev)
            w[:, 0, 0, :conv.weight.shape[1]] = conv.weight.detach().float()[:, :, 0]
            out += [w.contiguous(), conv.bias.detach().float().contiguous()]
        self._hipw = out
        self._hip_ld = ld
        return out

    def _forward_hip(self, coarse_logits, features):
        """the eval branch on emp_pr_* + emp_conv_bn_act_nhwc (D10): features stay NHWC, nothing but the point
        matrices (k points x 272 floats per image) is materialised"""
        from .. import _hip
        w = self._hip_weights()
        ld = self._hip_ld
        N, CF = features.shape[:2]
        logits = coarse_logits
        for _ in range(self.subdivision_steps):
            logits, unc = _hip.pr_upsample2x(logits)
            C, H, W = logits.shape[1:]
            k = min(H * W, self.subdivision_num_points)
            idx = _hip.pr_topk(unc, k)
            X0, X1 = _hip.pr_point_sample(features, coarse_logits, idx, H, W, ld)
            src, dst = X0, X1
            for li in range(len(self.point_head.fc_layers)):
                _hip.conv_bn_act_nhwc(_hip.as_pixels(src), w[2 * li], None, w[2 * li + 1], None, True,
                                      out=_hip.as_pixels(dst, CF))
                src, dst = dst, src
            pts = torch.empty((src.shape[0], C), dtype=torch.float32, device=src.device)
            _hip.conv_bn_act_nhwc(_hip.as_pixels(src), w[-2], None, w[-1], None, False, out=_hip.as_pixels(pts))
            _hip.pr_scatter(pts, idx, logits)
        return {'sem_seg_logits': logits}

    def forward(self, coarse_logits, features):
        assert not self.training, "only the inference path of PointRend is implemented"
        if (self.hip_ops and coarse_logits.is_cuda and coarse_logits.dtype == torch.float32
                and features.dtype == torch.float32 and features.stride(1) == 1
                and self.point_head.fc_layers[0][0].weight.shape[0] == features.shape[1]):
            return self._forward_hip(coarse_logits.contiguous(), features)
        features = features.contiguous()               # the library path works on NCHW-contiguous features
        logits = coarse_logits.clone()
        for _ in range(self.subdivision_steps):
            logits = F.interpolate(logits, scale_factor=2.0, mode='bilinear', align_corners=False)
            N, C, H, W = logits.shape
            k = min(H * W, self.subdivision_num_points)
            idx = torch.topk(self._uncertainty(logits).view(N, H * W), k=k, dim=1)[1]
            coords = torch.zeros(N, k, 2, dtype=torch.float, device=logits.device)
            coords[:, :, 0] = 0.5 / W + (idx % W).float() / float(W)
            coords[:, :, 1] = 0.5 / H + torch.div(idx, W, rounding_mode='floor').float() / float(H)
            pts = self.point_head(_point_sample(features, coords), _point_sample(coarse_logits, coords))
            logits = logits.reshape(N, C, H * W).scatter_(2, idx.unsqueeze(1).expand(-1, C, -1), pts).view(N, C, H, W)
        return {'sem_seg_logits': logits}


class PanopticDeepLabPR(PanopticDeepLab):
    """models/panoptic_deeplab.py:117-160 plus the 3-argument forward of the exported models that the Render
    engines call (`forward(x, render_steps, interpolate_ins)`, quantization/panoptic_deeplab.py:194-250)."""

    def __init__(self, num_fc=3, train_num_points=1024, oversample_ratio=3, importance_sample_ratio=0.75,
                 subdivision_steps=2, subdivision_num_points=8192, **kwargs):
        super().__init__(**kwargs)
        self.semantic_pr = PointRendSemSegHead(self.decoder_channels, self.num_classes, num_fc, train_num_points,
                                               oversample_ratio, importance_sample_ratio, subdivision_steps,
                                               subdivision_num_points)

    def forward(self, x, render_steps: int = 2, interpolate_ins: bool = True):
        pyramid = self.encoder(x)
        sem_x = self.semantic_decoder(pyramid)
        ins_x = sem_x if self.instance_decoder is None else self.instance_decoder(pyramid)
        self.semantic_pr.subdivision_steps = render_steps
        # PointRend: on the GPU the features stay NHWC (emp_pr_point_sample gathers whole pixels); the library path
        # makes them NCHW-contiguous itself
        sem = self.semantic_pr(self.semantic_head(sem_x).float().contiguous(), sem_x.float())
        ctr, off = self.ins_center(ins_x), self.ins_xy(ins_x)
        return {'sem_logits': sem['sem_seg_logits'],
                'ctr_hmp': self._up4(ctr) if interpolate_ins else ctr,
                'offsets': self._up4(off) if interpolate_ins else off}


# ----------------------------------------------------------------------------- deployment helpers
def synthesize_weights(model, scale_bn=True):
    """Deterministic synthetic weights (no trained checkpoints exist offline):
    per state-dict key, generator seeded with crc32(key); He-normal convs, BN gamma 1 +- 0.1,
    small beta / running_mean, running_var 1 + 0.1 * U[0,1)."""
    sd = model.state_dict()
    out = {}
    for key, t in sd.items():
        g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
        if key.endswith('num_batches_tracked'):
            out[key] = torch.zeros_like(t)
        elif key.endswith('running_var'):
            out[key] = 1 + 0.1 * torch.rand(t.shape, generator=g)
        elif key.endswith('running_mean'):
            out[key] = 0.1 * torch.randn(t.shape, generator=g)
        elif t.dim() in (3, 4):
            fan_in = t[0].numel()
            out[key] = torch.randn(t.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif key.endswith('weight'):
            out[key] = 1 + 0.1 * torch.randn(t.shape, generator=g)
            # last BatchNorm of a residual branch: small gamma (as zero-init-residual training leaves it), so that
            # activations stay O(1) through deep encoders instead of growing block after block
            if key.endswith(('bn3.weight', 'bottleneck.c.1.weight')) or (key.endswith('bn2.weight') and 'layer' in key
                                                                          and key.replace('bn2', 'bn3') not in sd):
                out[key] = 0.25 * out[key]
        else:
            out[key] = 0.05 * torch.randn(t.shape, generator=g)
    model.load_state_dict(out, strict=True)
    return model


def fuse_bn_act(model):
    """Swap every eval-mode BatchNorm2d (and the ReLU / residual add that follows it) for FusedBNAct.
    Patterns: ResNet blocks and stem (handled in their forward), Sequential[..., BatchNorm2d, ReLU, ...]."""
    for m in model.modules():
        if isinstance(m, (_Basic, _Bottleneck)):
            last = 'bn2' if isinstance(m, _Basic) else 'bn3'
            for name in ('bn1', 'bn2', 'bn3'):
                if hasattr(m, name) and isinstance(getattr(m, name), nn.BatchNorm2d):
                    setattr(m, name, FusedBNAct(getattr(m, name), relu=True))
            assert isinstance(getattr(m, last), FusedBNAct)
        elif isinstance(m, ResNetEncoder):
            m.bn1 = FusedBNAct(m.bn1, relu=True)
        elif isinstance(m, nn.Sequential):
            for i, child in enumerate(list(m)):
                if isinstance(child, nn.BatchNorm2d) and child.num_features % 4 == 0:
                    nxt = m[i + 1] if i + 1 < len(m) else None
                    relu = isinstance(nxt, nn.ReLU)
                    m[i] = FusedBNAct(child, relu=relu)
                    if relu:
                        m[i + 1] = nn.Identity()
    return model


def pair_conv_bn(model):
    """After fuse_bn_act: wrap every (Conv2d, FusedBNAct) producer/epilogue pair into one FusedConvBNAct call site
    (ResNet blocks, Sequential[Conv2d, FusedBNAct, ...], and the pointwise conv of a separable conv + its BN)."""
    for m in list(model.modules()):
        if isinstance(m, (_Basic, _Bottleneck)):
            for cn, bn_name in (('conv1', 'bn1'), ('conv2', 'bn2'), ('conv3', 'bn3')):
                if hasattr(m, cn) and isinstance(getattr(m, cn), nn.Conv2d) and isinstance(getattr(m, bn_name), FusedBNAct):
                    setattr(m, cn, FusedConvBNAct(getattr(m, cn), getattr(m, bn_name)))
                    setattr(m, bn_name, nn.Identity())
        elif isinstance(m, nn.Sequential):
            for i in range(len(m) - 1):
                if isinstance(m[i], nn.Conv2d) and isinstance(m[i + 1], FusedBNAct):
                    m[i] = FusedConvBNAct(m[i], m[i + 1])
                    m[i + 1] = nn.Identity()
                elif (isinstance(m[i], SeparableConv2d) and isinstance(m[i + 1], FusedBNAct)
                      and isinstance(m[i].sepconv[1], nn.Conv2d)):
                    m[i].sepconv[1] = FusedConvBNAct(m[i].sepconv[1], m[i + 1])
                    m[i + 1] = nn.Identity()
    return model


@torch.no_grad()
def tune_fused_convs(model, example, reps=5, verbose=False, allow=None, model_args=()):
    """Pick, per FusedConvBNAct call site, the fastest of its implementations on the shapes `example` produces
    (isolated timing with HIP events).  `allow`: optional collection restricting the candidates (e.g.
    ('miopen', 'direct') to keep every convolution in the direct form: the Winograd forms are fp32 too but round
    differently -- F(2x2) about 2x, F(3x3) / F(4x4) about 10x the direct form's rounding error).
    model_args: extra positional arguments of the model's forward (PointRend models: render_steps, interpolate_ins).
    Returns {module name: (chosen, {impl: ms})}."""
    model(example, *model_args)                      # records the shapes (and lets MIOpen pick its kernels)
    report = {}
    for name, m in model.named_modules():
        if not isinstance(m, FusedConvBNAct) or m._seen is None:
            continue
        shape, has_res = m._seen
        x = torch.randn(shape, device=example.device).contiguous(memory_format=torch.channels_last)
        m.impl = 'miopen'
        res = torch.randn_like(m(x)) if has_res else None
        times = {}
        for impl in m.candidates(has_res, shape):
            if allow is not None and impl not in allow and impl != 'miopen' and not (impl == 'direct_sk' and 'direct' in allow):
                continue
            m.impl = impl
            for _ in range(2):
                m(x, res)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                m(x, res)
            e1.record()
            torch.cuda.synchronize()
            times[impl] = e0.elapsed_time(e1) / reps
        best = min(times, key=times.get)
        if best == 'miopen' and len(times) > 1:
            # a library kernel has to win by more than the timing noise: within 2 % the hand-written form is kept
            # (same speed, and its summation order -- hence its output -- does not depend on a find step)
            own = min((k for k in times if k != 'miopen'), key=times.get)
            if times[own] <= 1.02 * times['miopen']:
                best = own
        m.impl, m.tuned = best, times
        m.release(best)
        report[name] = (best, {k: round(v, 4) for k, v in times.items()})
        if verbose:
            print(f"{name:55s} {shape} res={has_res} -> {best} {report[name][1]}", flush=True)
    return report


def swap_depthwise(model):
    """Swap every stride-1 depthwise Conv2d (3x3 / 5x5, zero "same" padding) for DepthwiseConvNHWC, and the last
    1x1 convolution of every head for PointwiseOutNHWC."""
    for m in model.modules():
        if isinstance(m, PanopticDeepLabHead) and PointwiseOutNHWC.eligible(m.head[1]):
            m.head[1] = PointwiseOutNHWC(m.head[1])
        for name, child in list(m.named_children()):
            if DepthwiseConvNHWC.eligible(child):
                if isinstance(m, nn.Sequential):
                    m[int(name)] = DepthwiseConvNHWC(child)
                else:
                    setattr(m, name, DepthwiseConvNHWC(child))
    return model


def prepare_for_inference(model, device='cuda', dtype=torch.float32, channels_last=True, fuse=True):
    """eval(), move to the GPU, NHWC memory format (MIOpen's fast layout on gfx950), optional bf16/fp16
    weights, and (fp32 NHWC only) the fused BatchNorm+ReLU(+residual) epilogue and depthwise-convolution kernels.  fp32 is the default so
    that logits stay within the stated tolerance of the CPU reference."""
    model = model.eval().to(device)
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    if dtype != torch.float32:
        model = model.to(dtype)
    elif fuse and channels_last and torch.device(device).type == 'cuda':
        model = swap_depthwise(pair_conv_bn(fuse_bn_act(model)))
        for m in list(model.modules()):
            if hasattr(m, 'hip_ops'):
                m.hip_ops = True                  # emp_upsample_bilinear + concat buffers written in place
            if hasattr(m, 'fuse_for_inference'):
                m.fuse_for_inference()            # RegNet blocks: shortcut + ReLU and the squeeze-excite gate fused
    return model
