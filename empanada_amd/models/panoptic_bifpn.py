"""Panoptic-BiFPN with a RegNet (or ResNet) encoder.

State-dict compatible re-implementation of the reference's second model family (cited per class):
  encoder   RegNetX/Y (AnyNet stages, group conv, per-pixel squeeze-excite)   empanada/models/encoders/regnet.py:37-316
  BiFPN     top-down + bottom-up weighted feature fusion, P6/P7 by max-pool   empanada/models/decoders/bifpn.py:17-198
  decoder   transposed-conv upsampling ladder + 5x5 separable fusion          empanada/models/decoders/bifpn.py:200-236
  model     PanopticBiFPN / PanopticBiFPNPR                                   empanada/models/panoptic_bifpn.py:22-172
Two reference behaviours are kept because they are visible in checkpoints / outputs:
  * every TopDownFPN / BottomUpFPN registers ONE conv block under several names (`after_combines.0/1/2...`,
    bifpn.py:34-42,90-98): the parameters are aliased, the state dict lists them once per name;
  * SqueezeExcite pools with AvgPool2d((1,1)) -- i.e. not at all -- so the gate is per pixel (blocks.py:37-53).
"""
from copy import deepcopy
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .panoptic_deeplab import (FusedConvBNAct, PanopticDeepLabHead, PointRendSemSegHead, SeparableConv2d, _RESNETS,
                               _conv_bn_act, _no_late_weights, _up_bilinear, resnet_encoder)

__all__ = ['PanopticBiFPN', 'PanopticBiFPNPR', 'regnet_encoder', 'REGNETS']

REGNETS = {
    'regnetx_6p4gf': dict(depth=17, w_0=184, w_a=60.83, w_m=2.07, group_w=56),
    'regnety_200mf': dict(depth=13, w_0=24, w_a=36.44, w_m=2.49, group_w=8),
    'regnety_800mf': dict(depth=14, w_0=56, w_a=38.84, w_m=2.4, group_w=16),
    'regnety_3p2gf': dict(depth=21, w_0=80, w_a=42.63, w_m=2.66, group_w=24),
    'regnety_4gf': dict(depth=22, w_0=96, w_a=31.41, w_m=2.24, group_w=64),
    'regnety_6p4gf': dict(depth=25, w_0=112, w_a=33.22, w_m=2.27, group_w=72, use_se=True),
    'regnety_8gf': dict(depth=17, w_0=192, w_a=76.82, w_m=2.19, group_w=56, use_se=True),
    'regnety_16gf': dict(depth=18, w_0=200, w_a=106.23, w_m=2.48, group_w=112, use_se=True),
}


class RegNetConfig:
    """Design-space parameters -> per-stage widths / depths / groups (regnet.py:171-258, arXiv 2003.13678 eq. 2-4)."""
    w_stem = 32
    bottle_ratio = 1

    def __init__(self, depth, w_0, w_a, w_m, group_w, q=8, use_se=False):
        assert w_a >= 0 and w_0 > 0 and w_m > 1 and w_0 % q == 0
        self.use_se = use_se
        self.strides = [2, 2, 2, 2]
        u = w_0 + np.arange(depth) * w_a
        s = np.round(np.log(u / w_0) / np.log(w_m))
        w = q * np.round(w_0 * np.power(w_m, s) / q).astype(int)
        w, d = np.unique(w, return_counts=True)
        assert len(w) == 4, "Bad parameters, only 4 stage networks allowed!"
        self.num_stages = 4
        self.depths = d.tolist()
        widths, groups = [], []
        for wi in w.tolist():
            w_b = int(max(1, wi * self.bottle_ratio))
            gw = int(min(group_w, w_b))
            w_b = max(gw, int(gw * round(w_b / gw)))
            widths.append(int(w_b / self.bottle_ratio))
            groups.append(w_b // gw)
        self.widths, self.groups = widths, groups


class SqueezeExcite(nn.Module):
    def __init__(self, nin):
        super().__init__()
        self.avg_pool = nn.AvgPool2d((1, 1))       # sic: a 1x1 average pool is the identity
        ns = nin // 4
        self.se = nn.Sequential(nn.Conv2d(nin, ns, 1, bias=True), nn.ReLU(inplace=True),
                                nn.Conv2d(ns, nin, 1, bias=True), nn.Sigmoid())

    def forward(self, x):
        return x * self.se(self.avg_pool(x))


class FusedSqueezeExcite(nn.Module):
    """Inference-only stand-in for SqueezeExcite on NHWC fp32 activations: two launches of the fused convolution
    kernel.  s = relu(W1 x + b1) (emp_conv_bn_act_nhwc, the squeeze width padded to a multiple of 16 with zero filters
    and zero bias, so the padding channels are exact zeros), then out = x * sigmoid(W2 s + b2) in the second launch's
    epilogue (relu == 2): the gate tensor, the sigmoid pass and the multiply pass never touch memory."""

    def __init__(self, se):
        super().__init__()
        c1, c2 = se.se[0], se.se[2]
        ns, nin = c1.out_channels, c1.in_channels
        nsp = -(-ns // 16) * 16
        w1 = torch.zeros(nsp, 1, 1, nin, dtype=torch.float32, device=c1.weight.device)
        w1[:ns] = c1.weight.detach().float().permute(0, 2, 3, 1)
        b1 = torch.zeros(nsp, dtype=torch.float32, device=c1.weight.device)
        b1[:ns] = c1.bias.detach().float()
        w2 = torch.zeros(nin, 1, 1, nsp, dtype=torch.float32, device=c1.weight.device)
        w2[..., :ns] = c2.weight.detach().float().permute(0, 2, 3, 1)
        for name, t in (('w1', w1), ('b1', b1), ('w2', w2), ('b2', c2.bias.detach().float().clone())):
            self.register_buffer(name, t.contiguous(), persistent=False)
        self._register_load_state_dict_pre_hook(_no_late_weights)

    @staticmethod
    def eligible(se):
        return (isinstance(se, SqueezeExcite) and se.se[0].in_channels % 16 == 0 and se.se[0].bias is not None
                and se.se[2].bias is not None and se.se[0].weight.dtype == torch.float32)

    def forward(self, x):
        from .. import _hip
        if not (x.is_cuda and x.dtype == torch.float32):
            raise RuntimeError("FusedSqueezeExcite needs fp32 CUDA activations")
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        s = _hip.conv_bn_act_nhwc(x, self.w1, None, self.b1, None, True)
        return _hip.conv_bn_act_nhwc(s, self.w2, None, self.b2, x, 'gate')


class Resample2d(nn.Module):
    """1x1 conv + BN when channels or stride change, identity otherwise (blocks.py:55-75)"""

    def __init__(self, nin, nout, stride=1):
        super().__init__()
        self.conv = _conv_bn_act(nin, nout, 1, stride=stride, act=False) if (nin != nout or stride > 1) else nn.Identity()

    def forward(self, x):
        return self.conv(x)


class _RegBottleneck(nn.Module):
    def __init__(self, w_in, w_out, groups, stride, use_se):
        super().__init__()
        self.a = _conv_bn_act(w_in, w_out, 1)
        self.b = _conv_bn_act(w_out, w_out, 3, stride=stride, groups=groups)
        if use_se:
            self.se = SqueezeExcite(w_out)
        self.c = _conv_bn_act(w_out, w_out, 1, act=False)
        self.use_se = use_se

    def forward(self, x):
        x = self.b(self.a(x))
        if self.use_se:
            x = self.se(x)
        return self.c(x)


class _RegBlock(nn.Module):
    def __init__(self, w_in, w_out, groups=1, stride=1, use_se=False):
        super().__init__()
        self.bottleneck = _RegBottleneck(w_in, w_out, groups, stride, use_se)
        self.downsample = Resample2d(w_in, w_out, stride=stride)
        self.act = nn.ReLU(inplace=True)
        self.fused_tail = False

    def fuse_for_inference(self):
        """After pair_conv_bn: the block's shortcut add + ReLU move into the epilogue of the last 1x1 convolution
        (residual operand of FusedConvBNAct) and the squeeze-excite becomes FusedSqueezeExcite."""
        b = self.bottleneck
        if isinstance(b.c[0], FusedConvBNAct) and not self.fused_tail:
            b.c[0].bn.relu = True
            self.fused_tail = True
            if b.use_se and FusedSqueezeExcite.eligible(b.se):
                b.se = FusedSqueezeExcite(b.se)

    def forward(self, x):
        if self.fused_tail:
            b = self.bottleneck
            y = b.b(b.a(x))
            if b.use_se:
                y = b.se(y)
            return b.c[0](y, self.downsample(x))
        return self.act(self.downsample(x) + self.bottleneck(x))


class _Stem(nn.Module):
    def __init__(self, w_in, w_out):
        super().__init__()
        self.cbr = _conv_bn_act(w_in, w_out, 3, stride=2)

    def forward(self, x):
        return self.cbr(x)


class RegNetEncoder(nn.Module):
    """returns [stem 1/2, stage1 1/4, stage2 1/8, stage3 1/16, stage4 1/32] (regnet.py:163-169)"""

    def __init__(self, cfg, im_channels=1):
        super().__init__()
        self.cfg = cfg
        self.stem = _Stem(im_channels, cfg.w_stem)
        w_ins = [cfg.w_stem] + cfg.widths[:-1]
        for i in range(cfg.num_stages):
            stage = nn.Sequential()
            for j in range(cfg.depths[i]):
                stage.add_module(f'block{j + 1}', _RegBlock(w_ins[i] if j == 0 else cfg.widths[i], cfg.widths[i],
                                                            cfg.groups[i], cfg.strides[i] if j == 0 else 1, cfg.use_se))
            self.add_module(f'stage{i + 1}', stage)

    def forward(self, x):
        feats = []
        for layer in self.children():
            x = layer(x)
            feats.append(x)
        return feats


def regnet_encoder(name):
    return RegNetEncoder(RegNetConfig(**REGNETS[name]))


def _fpn_conv_block(fpn_dim, depthwise):
    if depthwise:
        return nn.Sequential(SeparableConv2d(fpn_dim, fpn_dim, 3, 1, bias=False), nn.BatchNorm2d(fpn_dim),
                             nn.SiLU(inplace=True))
    return _conv_bn_act(fpn_dim, fpn_dim, 3)


def _fusion_weights(w, eps):
    w = F.relu(w)
    return w / (w.sum() + eps)


class TopDownFPN(nn.Module):
    """bifpn.py:17-71"""

    def __init__(self, pyramid_nins, fpn_dim, depthwise=True):
        super().__init__()
        self.resamplings = nn.ModuleList([Resample2d(nin, fpn_dim) for nin in pyramid_nins])
        block = _fpn_conv_block(fpn_dim, depthwise)
        self.after_combines = nn.ModuleList([block for _ in pyramid_nins])     # one module, many names
        self.weights = nn.Parameter(torch.ones(len(pyramid_nins) + 1), requires_grad=True)
        self.eps = 1e-4

    def forward(self, pyramid: List[torch.Tensor]):
        w = _fusion_weights(self.weights, self.eps)
        out = [pyramid[0]]
        for i, (resample, combine) in enumerate(zip(self.resamplings, self.after_combines)):
            up = F.interpolate(out[-1], scale_factor=2.0, mode='nearest')
            fused = (w[i] * up + w[i + 1] * resample(pyramid[i + 1])) / (w[i] + w[i + 1] + self.eps)
            out.append(combine(fused))
        return out


class BottomUpFPN(nn.Module):
    """bifpn.py:73-134"""

    def __init__(self, pyramid_nins, fpn_dim, depthwise=True):
        super().__init__()
        self.resamplings = nn.ModuleList([Resample2d(nin, fpn_dim) for nin in pyramid_nins])
        block = _fpn_conv_block(fpn_dim, depthwise)
        self.after_combines = nn.ModuleList([block for _ in pyramid_nins])
        self.weights = nn.Parameter(torch.ones(len(pyramid_nins) + 1), requires_grad=True)
        self.eps = 1e-4

    def forward(self, pyramid: List[torch.Tensor], top_down: List[torch.Tensor]):
        w = _fusion_weights(self.weights, self.eps)
        out = [top_down[0]]
        n = len(self.resamplings)
        for i, (resample, combine) in enumerate(zip(self.resamplings, self.after_combines)):
            down = F.max_pool2d(out[-1], 3, stride=2, padding=1)
            lateral = resample(pyramid[i])
            if i < n - 1:
                fused = (w[i] * down + w[i + 1] * lateral + w[i + 2] * top_down[i + 1]) / (w[i] + w[i + 1] + w[i + 2] + self.eps)
            else:
                fused = (w[i] * down + w[i + 1] * lateral) / (w[i] + w[i + 1] + self.eps)
            out.append(combine(fused))
        return out


class BiFPNLayer(nn.Module):
    """bifpn.py:136-156"""

    def __init__(self, pyramid_nins, fpn_dim, depthwise=True):
        super().__init__()
        self.top_down_fpn = TopDownFPN(pyramid_nins[::-1][1:], fpn_dim, depthwise)
        self.bottom_up_fpn = BottomUpFPN(pyramid_nins[1:], fpn_dim, depthwise)

    def forward(self, pyramid: List[torch.Tensor]):
        td = self.top_down_fpn(pyramid[::-1])
        return self.bottom_up_fpn(pyramid[1:], td[::-1])


class BiFPN(nn.Module):
    """bifpn.py:158-198: P6/P7 from the last stage by 1x1 resample + 3x3/2 max-pools, then `num_layers` BiFPN layers"""

    def __init__(self, pyramid_nins, fpn_dim, num_layers=3, depthwise=True):
        super().__init__()
        self.p6_resample = Resample2d(pyramid_nins[-1], fpn_dim)
        nins = list(pyramid_nins) + [fpn_dim, fpn_dim]
        self.bifpns = nn.ModuleList([BiFPNLayer(nins if i == 0 else len(nins) * [fpn_dim], fpn_dim, depthwise)
                                     for i in range(num_layers)])

    def forward(self, pyramid: List[torch.Tensor]):
        p6 = F.max_pool2d(self.p6_resample(pyramid[-1]), 3, stride=2, padding=1)
        p7 = F.max_pool2d(p6, 3, stride=2, padding=1)
        pyramid = list(pyramid) + [p6, p7]
        for layer in self.bifpns:
            pyramid = layer(pyramid)
        return pyramid


class BiFPNDecoder(nn.Module):
    """bifpn.py:200-236"""

    def __init__(self, fpn_dim, n_fpn_scales=5):
        super().__init__()
        self.n_fpn_scales = n_fpn_scales
        self.upsamplings = nn.ModuleList([
            nn.Sequential(nn.ConvTranspose2d(fpn_dim if i == 0 else 2 * fpn_dim, fpn_dim, 2, stride=2, bias=False),
                          nn.BatchNorm2d(fpn_dim), nn.ReLU(inplace=True)) for i in range(n_fpn_scales)])
        self.fusion = nn.Sequential(SeparableConv2d(2 * fpn_dim, fpn_dim, 5, 1, bias=False), nn.BatchNorm2d(fpn_dim),
                                    nn.ReLU(inplace=True))

    def forward(self, feats: List[torch.Tensor]):
        assert len(feats) == self.n_fpn_scales + 1
        x = feats[0]
        for up, skip in zip(self.upsamplings, feats[1:]):
            x = torch.cat([up(x), skip], dim=1)
        return self.fusion(x)


class PanopticBiFPN(nn.Module):
    """panoptic_bifpn.py:22-126.  forward(x (N,1,H,W)) -> {'sem_logits','ctr_hmp','offsets'} at input resolution."""

    def __init__(self, encoder='regnety_6p4gf', num_classes=1, fpn_dim=160, fpn_layers=3, ins_decoder=False,
                 depthwise=True, **kwargs):
        super().__init__()
        assert encoder in REGNETS or encoder in _RESNETS, f'Invalid encoder name {encoder}'
        # the reference builds the encoder with default arguments: output stride 32 (panoptic_bifpn.py:38)
        self.encoder = regnet_encoder(encoder) if encoder in REGNETS else resnet_encoder(encoder, output_stride=32)
        widths = [int(w) for w in self.encoder.cfg.widths]
        self.p2_resample = Resample2d(widths[0], fpn_dim)
        self.num_classes = num_classes
        self.fpn_dim = fpn_dim
        self.semantic_fpn = BiFPN(deepcopy(widths[1:]), fpn_dim, fpn_layers, depthwise)
        self.semantic_decoder = BiFPNDecoder(fpn_dim)
        if ins_decoder:
            self.instance_fpn = BiFPN(deepcopy(widths[1:]), fpn_dim, fpn_layers, depthwise)
            self.instance_decoder = BiFPNDecoder(fpn_dim)
        else:
            self.instance_fpn = None
        self.semantic_head = PanopticDeepLabHead(fpn_dim, num_classes)
        self.ins_center = PanopticDeepLabHead(fpn_dim, 1)
        self.ins_xy = PanopticDeepLabHead(fpn_dim, 2)

    hip_ops = False

    def _up4(self, x):
        return _up_bilinear(x, (4 * x.shape[2], 4 * x.shape[3]), self.hip_ops)

    def _features(self, x):
        pyramid = self.encoder(x)
        p2 = self.p2_resample(pyramid[1])
        sem_x = self.semantic_decoder(([p2] + self.semantic_fpn(pyramid[2:]))[::-1])
        if self.instance_fpn is not None:
            ins_x = self.instance_decoder(([p2] + self.instance_fpn(pyramid[2:]))[::-1])
        else:
            ins_x = sem_x
        return sem_x, ins_x

    def forward(self, x):
        sem_x, ins_x = self._features(x)
        return {'sem_logits': self._up4(self.semantic_head(sem_x)), 'ctr_hmp': self._up4(self.ins_center(ins_x)),
                'offsets': self._up4(self.ins_xy(ins_x))}


class PanopticBiFPNPR(PanopticBiFPN):
    """panoptic_bifpn.py:128-172 + the exported 3-argument forward (quantization/panoptic_bifpn.py:105-161)."""

    def __init__(self, num_fc=3, train_num_points=1024, oversample_ratio=3, importance_sample_ratio=0.75,
                 subdivision_steps=2, subdivision_num_points=8192, **kwargs):
        super().__init__(**kwargs)
        self.semantic_pr = PointRendSemSegHead(self.fpn_dim, self.num_classes, num_fc, train_num_points,
                                               oversample_ratio, importance_sample_ratio, subdivision_steps,
                                               subdivision_num_points)

    def forward(self, x, render_steps: int = 2, interpolate_ins: bool = True):
        sem_x, ins_x = self._features(x)
        self.semantic_pr.subdivision_steps = render_steps
        sem = self.semantic_pr(self.semantic_head(sem_x).float().contiguous(), sem_x.float())
        ctr, off = self.ins_center(ins_x), self.ins_xy(ins_x)
        return {'sem_logits': sem['sem_seg_logits'], 'ctr_hmp': self._up4(ctr) if interpolate_ins else ctr,
                'offsets': self._up4(off) if interpolate_ins else off}
