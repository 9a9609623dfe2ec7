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
