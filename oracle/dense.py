"""CPU oracle for the dense-path helper kernels (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

dwconv_nhwc: depthwise half of SeparableConv2d (empanada/models/blocks.py:15-33) with the summation order of
include/emp_hip.h (D2) fixed, plain C (oracle/c/oracle_kernels.c).  Pinned against torch's conv2d (the library
call the reference makes) within fp32 rounding; the exact order itself is this framework's contract.
"""
import ctypes

import numpy as np

from ._clib import lib


def dwconv_nhwc(x_nhwc, w_kkc, bias=None):
    """x (N,H,W,C) fp32, w (k*k, C) fp32, bias (C) or None -> (N,H,W,C) fp32"""
    x = np.ascontiguousarray(x_nhwc, dtype=np.float32)
    w = np.ascontiguousarray(w_kkc, dtype=np.float32)
    N, H, W, C = x.shape
    k = int(round(np.sqrt(w.shape[0])))
    assert k * k == w.shape[0] and w.shape[1] == C
    y = np.empty_like(x)
    f32p = ctypes.POINTER(ctypes.c_float)
    b = None
    if bias is not None:
        b = np.ascontiguousarray(bias, dtype=np.float32)
    lib().emp_oracle_dwconv_nhwc(x.ctypes.data_as(f32p), w.ctypes.data_as(f32p),
                                 b.ctypes.data_as(f32p) if b is not None else None, N, H, W, C, k,
                                 y.ctypes.data_as(f32p))
    return y


def upsample_bilinear(x, size):
    """bilinear, align_corners=True, (N,C,h,w) fp32 -> (N,C,H,W): restatement of what the reference's
    F.interpolate(..., mode='bilinear', align_corners=True) computes (empanada/models/panoptic_deeplab.py:100-113,
    decoders/panoptic_deeplab.py:70-78) with the operation order of include/emp_hip.h (D3) fixed; every numpy
    fp32 operation is one rounding.  Pinned against torch's own interpolate within 1e-6 in the GPU tests."""
    x = np.asarray(x, dtype=np.float32)
    N, C, h, w = x.shape
    H, W = size
    f = np.float32

    def src(n_in, n_out):
        r = f(n_in - 1) / f(n_out - 1) if n_out > 1 else f(0)
        s = (r * np.arange(n_out, dtype=np.float32)).astype(np.float32)
        i0 = np.minimum(s.astype(np.int64), n_in - 1)
        i1 = i0 + (i0 < n_in - 1)
        return i0, i1, (s - i0.astype(np.float32)).astype(np.float32)

    y0, y1, ly = src(h, H)
    x0, x1, lx = src(w, W)
    lx0, ly0 = (f(1) - lx).astype(np.float32), (f(1) - ly).astype(np.float32)
    v00, v01 = x[:, :, y0][:, :, :, x0], x[:, :, y0][:, :, :, x1]
    v10, v11 = x[:, :, y1][:, :, :, x0], x[:, :, y1][:, :, :, x1]
    top = (lx0 * v00).astype(np.float32) + (lx * v01).astype(np.float32)
    bot = (lx0 * v10).astype(np.float32) + (lx * v11).astype(np.float32)
    lyc, ly0c = ly[:, None], ly0[:, None]
    return ((ly0c * top).astype(np.float32) + (lyc * bot).astype(np.float32)).astype(np.float32)


def conv_bn_act_nhwc(x_nhwc, w_okkc, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0, dil=1):
    """x (N,H,W,Cin), w (Cout,KH,KW,Cin), per-channel scale/shift, residual (N,OH,OW,Cout) -> (N,OH,OW,Cout),
    all fp32; summation order of include/emp_hip.h (D4).  Plain C (oracle/c/oracle_kernels.c)."""
    x = np.ascontiguousarray(x_nhwc, dtype=np.float32)
    w = np.ascontiguousarray(w_okkc, dtype=np.float32)
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    y = np.empty((N, OH, OW, Cout), dtype=np.float32)
    f32p = ctypes.POINTER(ctypes.c_float)

    def ptr(a):
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float32)
        return a, a.ctypes.data_as(f32p)

    keep = [ptr(a) for a in (scale, shift, residual)]
    lib().emp_oracle_conv_bn_act_nhwc(x.ctypes.data_as(f32p), w.ctypes.data_as(f32p), keep[0][1], keep[1][1],
                                      keep[2][1], int(bool(relu)), N, H, W, Cin, Cout, KH, KW, stride, pad, dil,
                                      y.ctypes.data_as(f32p))
    return y
