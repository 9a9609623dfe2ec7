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


def conv_bn_act_nhwc(x_nhwc, w_okkc, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0, dil=1,
                     slab=32):
    """x (N,H,W,Cin), w (Cout,KH,KW,Cin), per-channel scale/shift, residual (N,OH,OW,Cout) -> (N,OH,OW,Cout),
    all fp32; summation order of include/emp_hip.h (D4) for the K-slab the kernel uses (emp_conv_k_slab).  Plain C
    (oracle/c/oracle_kernels.c)."""
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
    act = 2 if relu == 'gate' else int(bool(relu))       # 'gate': y = residual * sigmoid(acc * scale + shift)
    lib().emp_oracle_conv_bn_act_nhwc(x.ctypes.data_as(f32p), w.ctypes.data_as(f32p), keep[0][1], keep[1][1],
                                      keep[2][1], act, N, H, W, Cin, Cout, KH, KW, stride, pad, dil,
                                      int(slab), y.ctypes.data_as(f32p))
    return y


def conv_splitk_bn_act_nhwc(x_nhwc, w_okkc, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0, dil=1,
                            k_splits=2):
    """The split-K form of the convolution above (include/emp_hip.h, D4c: emp_conv_splitk_bn_act_nhwc): partial fmaf
    chains over k_splits ranges of 32-channel slabs, added in ascending order, then the epilogue.  Plain C."""
    x = np.ascontiguousarray(x_nhwc, dtype=np.float32)
    w = np.ascontiguousarray(w_okkc, dtype=np.float32)
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    OH = (H + 2 * pad - dil * (KH - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (KW - 1) - 1) // stride + 1
    y = np.empty((N, OH, OW, Cout), dtype=np.float32)
    f32p = ctypes.POINTER(ctypes.c_float)
    keep = [None if a is None else np.ascontiguousarray(a, dtype=np.float32) for a in (scale, shift, residual)]
    ptrs = [None if a is None else a.ctypes.data_as(f32p) for a in keep]
    lib().emp_oracle_conv_splitk_bn_act_nhwc(x.ctypes.data_as(f32p), w.ctypes.data_as(f32p), ptrs[0], ptrs[1], ptrs[2],
                                             int(bool(relu)), N, H, W, Cin, Cout, KH, KW, stride, pad, dil,
                                             int(k_splits), y.ctypes.data_as(f32p))
    return y


def gconv_chunk(group_w):
    """channels per K chunk of the grouped kernel (emp_gconv_chunk)"""
    return 24 if group_w % 24 == 0 else 16 if group_w % 16 == 0 else 8


def gconv3x3_bn_act_nhwc(x_nhwc, w_okkc, groups, scale=None, shift=None, relu=False, stride=1):
    """Grouped 3x3 convolution (padding 1) + affine + ReLU: x (N,H,W,C), w (C,3,3,C/groups), summation order of
    include/emp_hip.h (D8).  Restates the grouped Conv2d + BatchNorm2d + ReLU of the RegNet bottleneck
    (empanada/models/encoders/regnet.py:59-71, blocks.py:121-171); pinned against torch's conv2d(groups=G) within
    fp32 rounding in tests/test_oracle_dense.py.  Plain C (oracle/c/oracle_kernels.c)."""
    x = np.ascontiguousarray(x_nhwc, dtype=np.float32)
    w = np.ascontiguousarray(w_okkc, dtype=np.float32)
    N, H, W, C = x.shape
    GW = C // groups
    assert w.shape == (C, 3, 3, GW) and GW * groups == C and GW % 8 == 0
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = np.empty((N, OH, OW, C), dtype=np.float32)
    f32p = ctypes.POINTER(ctypes.c_float)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
    lib().emp_oracle_gconv3x3_bn_act_nhwc(x.ctypes.data_as(f32p), w.ctypes.data_as(f32p),
                                          None if sc is None else sc.ctypes.data_as(f32p),
                                          None if sh is None else sh.ctypes.data_as(f32p), int(bool(relu)),
                                          N, H, W, groups, GW, stride, gconv_chunk(GW), y.ctypes.data_as(f32p))
    return y


def wino_filter_transform(w_oihw):
    """U (16, Cout, Cin) = G g G^T in the operation order of empanada_amd._hip.wino_filter_transform (numpy fp32)."""
    g = np.asarray(w_oihw, dtype=np.float32)
    h = np.float32(0.5)

    def comb(a, b, c):
        return [a, (h * ((a + b) + c)).astype(np.float32), (h * ((a - b) + c)).astype(np.float32), c]

    rows = comb(g[:, :, 0, :], g[:, :, 1, :], g[:, :, 2, :])
    U = []
    for r in rows:
        U.extend(comb(r[:, :, 0], r[:, :, 1], r[:, :, 2]))
    return np.ascontiguousarray(np.stack(U, axis=0), dtype=np.float32)


def wino_conv_bn_act(x_nhwc, w_oihw, tiles, dil, scale=None, shift=None, relu=False, slab=32):
    """Winograd F(2x2,3x3) convolution exactly as include/emp_hip.h (D5) specifies it: input transform (numpy fp32,
    one rounding per add), 16 GEMMs through the C fma-chain oracle, output transform + epilogue.  Pinned against
    torch's conv2d within a stated tolerance in the GPU tests (the Winograd form is this framework's choice; the
    reference calls the backend's convolution)."""
    x = np.asarray(x_nhwc, dtype=np.float32)
    N, H, W, C = x.shape
    U = wino_filter_transform(w_oihw)
    Cout = U.shape[1]
    T = len(tiles)
    xp = np.zeros((N, H + 6 * dil + 8, W + 6 * dil + 8, C), dtype=np.float32)       # generous zero border
    off = 3 * dil + 4
    xp[:, off:off + H, off:off + W] = x
    n, by, bx = tiles[:, 0], tiles[:, 1] + off, tiles[:, 2] + off
    d = [[xp[n, by + a * dil, bx + b * dil] for b in range(4)] for a in range(4)]    # each (T, C)
    t = [[d[a][0] - d[a][2], d[a][1] + d[a][2], d[a][2] - d[a][1], d[a][1] - d[a][3]] for a in range(4)]
    V = np.empty((16, T, C), dtype=np.float32)
    for v in range(4):
        V[0 * 4 + v] = t[0][v] - t[2][v]
        V[1 * 4 + v] = t[1][v] + t[2][v]
        V[2 * 4 + v] = t[2][v] - t[1][v]
        V[3 * 4 + v] = t[1][v] - t[3][v]
    M = np.empty((16, T, Cout), dtype=np.float32)
    for p in range(16):          # GEMM as a 1x1 convolution over a (1, 1, T, C) image
        M[p] = conv_bn_act_nhwc(V[p][None, None], U[p][:, None, None, :], slab=slab)[0, 0]
    m = [[M[a * 4 + b] for b in range(4)] for a in range(4)]
    s = [[(m[0][b] + m[1][b]) + m[2][b] for b in range(4)], [(m[1][b] - m[2][b]) - m[3][b] for b in range(4)]]
    out = np.zeros((N, H, W, Cout), dtype=np.float32)
    for a in range(2):
        yv = [(s[a][0] + s[a][1]) + s[a][2], (s[a][1] - s[a][2]) - s[a][3]]
        for b in range(2):
            v = yv[b]
            if scale is not None:
                v = (v * np.asarray(scale, dtype=np.float32)).astype(np.float32)
            if shift is not None:
                v = (v + np.asarray(shift, dtype=np.float32)).astype(np.float32)
            if relu:
                v = np.maximum(v, np.float32(0))
            yy, xx = tiles[:, 1] + dil + a * dil, tiles[:, 2] + dil + b * dil
            ok = (yy < H) & (xx < W)
            out[tiles[ok, 0], yy[ok], xx[ok]] = v[ok]
    return out


def pointwise_out_nhwc(x_nhwc, w, bias=None):
    """1x1 convolution to a few channels with the summation order of include/emp_hip.h (D6): per lane an fma chain
    over its channel groups, butterfly sum over 64 lanes, + bias.  numpy fp32; the fma is emulated exactly in fp64
    (a product of two fp32 values is exact in fp64, and fp64 -> fp32 rounding of (exact product + fp32 addend) equals
    fmaf whenever the fp64 sum is exact or its double rounding is innocuous -- checked against the GPU bit for bit)."""
    x = np.asarray(x_nhwc, dtype=np.float32)
    w = np.asarray(w, dtype=np.float32)
    N, H, W, C = x.shape
    Cout = w.shape[0]
    C4 = C // 4
    P = N * H * W
    xf = x.reshape(P, C)
    out = np.empty((N, Cout, H * W), dtype=np.float32)
    for co in range(Cout):
        part = np.zeros((P, 64), dtype=np.float32)
        for g in range((C4 + 63) // 64):
            for l in range(64):
                c4 = l + 64 * g
                if c4 >= C4:
                    continue
                for e in range(4):
                    c = 4 * c4 + e
                    part[:, l] = (xf[:, c].astype(np.float64) * np.float64(w[co, c]) + part[:, l].astype(np.float64)).astype(np.float32)
        s = part
        for o in (32, 16, 8, 4, 2, 1):
            s = (s + s[:, np.arange(64) ^ o]).astype(np.float32)
        v = s[:, 0]
        if bias is not None:
            v = (v + np.float32(bias[co])).astype(np.float32)
        out[:, co] = v.reshape(N, H * W)
    return out.reshape(N, Cout, H, W)


def bn_relu_maxpool_nhwc(x_nhwc, scale, shift):
    """max over 3x3 / stride 2 / pad 1 windows of max(x*scale + shift, 0)  (emp_hip.h D7), numpy fp32"""
    x = np.asarray(x_nhwc, dtype=np.float32)
    N, H, W, C = x.shape
    a = np.maximum((x * np.asarray(scale, np.float32)).astype(np.float32) + np.asarray(shift, np.float32), np.float32(0))
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    pad = np.full((N, H + 2, W + 2, C), -np.inf, dtype=np.float32)
    pad[:, 1:H + 1, 1:W + 1] = a
    out = np.full((N, OH, OW, C), -np.inf, dtype=np.float32)
    for dy in range(3):
        for dx in range(3):
            out = np.maximum(out, pad[:, dy:dy + 2 * OH:2, dx:dx + 2 * OW:2][:, :OH, :OW])
    return out


def logits_to_prob(logits):
    """logits_to_prob (empanada/inference/engines.py:22-30) in the operation order of include/emp_hip.h (D2), numpy fp32
    (np.exp on float32 = the host libm's expf: the HIP kernel's device expf may differ in the last place)"""
    x = np.asarray(logits, dtype=np.float32)
    one = np.float32(1)
    if x.shape[1] == 1:
        return (one / (one + np.exp(-x))).astype(np.float32)
    m = x.max(axis=1, keepdims=True)
    e = np.exp((x - m).astype(np.float32)).astype(np.float32)
    s = np.zeros_like(m)
    for c in range(x.shape[1]):
        s = (s + e[:, c:c + 1]).astype(np.float32)
    return (e / s).astype(np.float32)


def stem_conv7_bn_relu_maxpool(x_nhw, w_tc, scale, shift):
    """The ResNet stem (encoders/resnet.py:186-188,217-222: conv1 7x7 / 2 -> bn1 -> relu -> maxpool 3x3 / 2) on a
    one-channel image x (N,H,W): convolution as one fmaf chain per output over the 49 taps in raster order (plain C),
    then the D7 epilogue above.  w_tc: (49, Cout) = conv1.weight[:, 0].reshape(Cout, 49).T.  Order of include/emp_hip.h
    (D9); pinned against torch's conv2d + max_pool2d within fp32 rounding in tests/test_oracle_dense.py."""
    x = np.ascontiguousarray(x_nhw, dtype=np.float32)
    w = np.ascontiguousarray(w_tc, dtype=np.float32)
    N, H, W = x.shape
    Cout = w.shape[1]
    assert w.shape[0] == 49
    conv = np.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cout), dtype=np.float32)
    f32p = ctypes.POINTER(ctypes.c_float)
    lib().emp_oracle_conv7s2_c1(x.ctypes.data_as(f32p), w.ctypes.data_as(f32p), N, H, W, Cout, conv.ctypes.data_as(f32p))
    return bn_relu_maxpool_nhwc(conv, scale, shift)


def wino4_filter_transform(w_oihw):
    """fp32(G g G^T) for F(4x4,3x3) in the elementwise fp64 order of empanada_amd._hip.wino4_filter_transform"""
    g = np.asarray(w_oihw, dtype=np.float64)

    def comb(a, b, c):
        return [a / 4.0, -((a + b) + c) / 6.0, ((b - a) - c) / 6.0, (a / 24.0 + b / 12.0) + c / 6.0,
                (a / 24.0 - b / 12.0) + c / 6.0, c]

    rows = comb(g[:, :, 0, :], g[:, :, 1, :], g[:, :, 2, :])
    U = []
    for r in rows:
        U.extend(comb(r[:, :, 0], r[:, :, 1], r[:, :, 2]))
    return np.ascontiguousarray(np.stack(U, axis=0).astype(np.float32))


def _bt4(d):
    f = np.float32
    return [(f(4) * d[0] - f(5) * d[2]) + d[4], (d[3] + d[4]) - f(4) * (d[1] + d[2]), (d[4] - d[3]) + f(4) * (d[1] - d[2]),
            (d[4] - d[2]) + f(2) * (d[3] - d[1]), (d[4] - d[2]) + f(2) * (d[1] - d[3]), (f(4) * d[1] - f(5) * d[3]) + d[5]]


def _at4(m):
    f = np.float32
    return [((m[0] + m[1]) + m[2]) + (m[3] + m[4]), (m[1] - m[2]) + f(2) * (m[3] - m[4]),
            (m[1] + m[2]) + f(4) * (m[3] + m[4]), ((m[1] - m[2]) + f(8) * (m[3] - m[4])) + m[5]]


def wino4_conv_bn_act(x_nhwc, w_oihw, tiles, dil, scale=None, shift=None, relu=False, slab=32):
    """Winograd F(4x4,3x3) exactly as include/emp_hip.h (D5b) specifies it (numpy fp32, one rounding per operation;
    GEMMs through the C fma-chain oracle)."""
    x = np.asarray(x_nhwc, dtype=np.float32)
    N, H, W, C = x.shape
    U = wino4_filter_transform(w_oihw)
    Cout = U.shape[1]
    T = len(tiles)
    off = 5 * dil + 8
    xp = np.zeros((N, H + 2 * off, W + 2 * off, C), dtype=np.float32)
    xp[:, off:off + H, off:off + W] = x
    n, by, bx = tiles[:, 0], tiles[:, 1] + off, tiles[:, 2] + off
    tt = [_bt4([xp[n, by + a * dil, bx + b * dil] for b in range(6)]) for a in range(6)]      # tt[a][v]
    V = np.empty((36, T, C), dtype=np.float32)
    for v in range(6):
        r = _bt4([tt[a][v] for a in range(6)])
        for u in range(6):
            V[u * 6 + v] = r[u]
    M = np.empty((36, T, Cout), dtype=np.float32)
    for p in range(36):
        M[p] = conv_bn_act_nhwc(V[p][None, None], U[p][:, None, None, :], slab=slab)[0, 0]
    s = [[None] * 6 for _ in range(4)]
    for b in range(6):
        r = _at4([M[a * 6 + b] for a in range(6)])
        for a in range(4):
            s[a][b] = r[a]
    out = np.zeros((N, H, W, Cout), dtype=np.float32)
    for a in range(4):
        yv = _at4(s[a])
        for b in range(4):
            v = yv[b]
            if scale is not None:
                v = (v * np.asarray(scale, dtype=np.float32)).astype(np.float32)
            if shift is not None:
                v = (v + np.asarray(shift, dtype=np.float32)).astype(np.float32)
            if relu:
                v = np.maximum(v, np.float32(0))
            yy, xx = tiles[:, 1] + dil + a * dil, tiles[:, 2] + dil + b * dil
            ok = (yy < H) & (xx < W)
            out[tiles[ok, 0], yy[ok], xx[ok]] = v[ok]
    return out


_W3_BT = [[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0], [0, 2, -1, -2, 1]]
_W3_AT = [[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 1]]
_W3_G = [[0.5, 0.0, 0.0], [-0.5, -0.5, -0.5], [-1.0 / 6, 1.0 / 6, -1.0 / 6], [1.0 / 6, 1.0 / 3, 2.0 / 3], [0.0, 0.0, 1.0]]


def _fold(mat, d):
    """r[u] = left fold over the non-zero entries of row u of fl(c * d[a]), fp32"""
    out = []
    for row in mat:
        acc = None
        for c, v in zip(row, d):
            if c != 0:
                term = (np.float32(c) * v).astype(np.float32)
                acc = term if acc is None else (acc + term).astype(np.float32)
        out.append(acc)
    return out


def wino3_filter_transform(w_oihw):
    g = np.asarray(w_oihw, dtype=np.float64)

    def comb(a, b, c):
        return [(G[0] * a + G[1] * b) + G[2] * c for G in _W3_G]

    rows = comb(g[:, :, 0, :], g[:, :, 1, :], g[:, :, 2, :])
    U = []
    for r in rows:
        U.extend(comb(r[:, :, 0], r[:, :, 1], r[:, :, 2]))
    return np.ascontiguousarray(np.stack(U, axis=0).astype(np.float32))


def wino3_conv_bn_act(x_nhwc, w_oihw, tiles, dil, scale=None, shift=None, relu=False, slab=32):
    """Winograd F(3x3,3x3) exactly as include/emp_hip.h (D5c) specifies it."""
    x = np.asarray(x_nhwc, dtype=np.float32)
    N, H, W, C = x.shape
    U = wino3_filter_transform(w_oihw)
    Cout = U.shape[1]
    T = len(tiles)
    off = 4 * dil + 8
    xp = np.zeros((N, H + 2 * off, W + 2 * off, C), dtype=np.float32)
    xp[:, off:off + H, off:off + W] = x
    n, by, bx = tiles[:, 0], tiles[:, 1] + off, tiles[:, 2] + off
    tt = [_fold(_W3_BT, [xp[n, by + a * dil, bx + b * dil] for b in range(5)]) for a in range(5)]
    V = np.empty((25, T, C), dtype=np.float32)
    for v in range(5):
        r = _fold(_W3_BT, [tt[a][v] for a in range(5)])
        for u in range(5):
            V[u * 5 + v] = r[u]
    M = np.empty((25, T, Cout), dtype=np.float32)
    for p in range(25):
        M[p] = conv_bn_act_nhwc(V[p][None, None], U[p][:, None, None, :], slab=slab)[0, 0]
    s = [[None] * 5 for _ in range(3)]
    for b in range(5):
        r = _fold(_W3_AT, [M[a * 5 + b] for a in range(5)])
        for a in range(3):
            s[a][b] = r[a]
    out = np.zeros((N, H, W, Cout), dtype=np.float32)
    for a in range(3):
        yv = _fold(_W3_AT, s[a])
        for b in range(3):
            v = yv[b]
            if scale is not None:
                v = (v * np.asarray(scale, dtype=np.float32)).astype(np.float32)
            if shift is not None:
                v = (v + np.asarray(shift, dtype=np.float32)).astype(np.float32)
            if relu:
                v = np.maximum(v, np.float32(0))
            yy, xx = tiles[:, 1] + dil + a * dil, tiles[:, 2] + dil + b * dil
            ok = (yy < H) & (xx < W)
            out[tiles[ok, 0], yy[ok], xx[ok]] = v[ok]
    return out


def conv_bn_act_proj_nhwc(x_nhwc, w_okkc, scale, shift, relu, proj_w, proj_b=None, stride=1, pad=0, dil=1, slab=32):
    """emp_conv_bn_act_proj_nhwc (emp_hip.h, D4 + D6): activation from the conv oracle, then per cout tile of 128 the
    lane partials ((v0 w0 + v1 w1) + v2 w2) + v3 w3, their left fold over the 32 lanes, the sum of the (at most two) tiles
    and the bias -- numpy fp32, one rounding per operation.  Returns planar (N, n, OH, OW)."""
    y = conv_bn_act_nhwc(x_nhwc, w_okkc, scale, shift, None, relu, stride, pad, dil, slab)     # (N, OH, OW, Cout)
    N, OH, OW, Cout = y.shape
    pw = np.asarray(proj_w, dtype=np.float32)
    n = pw.shape[0]
    P = N * OH * OW
    yf = y.reshape(P, Cout)
    out = np.zeros((P, n), dtype=np.float32)
    for q in range(n):
        total = None
        for t0 in range(0, Cout, 128):
            v = yf[:, t0:t0 + 128].reshape(P, 32, 4)
            w = pw[q, t0:t0 + 128].reshape(32, 4)
            pr = (v * w[None]).astype(np.float32)
            s = (((pr[:, :, 0] + pr[:, :, 1]).astype(np.float32) + pr[:, :, 2]).astype(np.float32) + pr[:, :, 3]).astype(np.float32)
            acc = s[:, 0]
            for j in range(1, 32):
                acc = (acc + s[:, j]).astype(np.float32)
            total = acc if total is None else (total + acc).astype(np.float32)
        if proj_b is not None:
            total = (total + np.float32(proj_b[q])).astype(np.float32)
        out[:, q] = total
    return np.ascontiguousarray(out.reshape(N, OH * OW, n).transpose(0, 2, 1)).reshape(N, n, OH, OW)


# ----------------------------------------------------------------------------- D10: PointRend subdivision step
def pr_upsample2x(x):
    """F.interpolate(scale_factor=2, bilinear, align_corners=False) + calculate_uncertainty
    (empanada/models/point_rend.py:62-79, 244-248), fp32 op for op as emp_pr_upsample2x documents it.
    x (N,C,h,w) -> (up (N,C,2h,2w), uncertainty (N, 4hw))."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    N, C, h, w = x.shape
    f = np.float32

    def src(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float32) + f(0.5)) * f(0.5) - f(0.5)
        s = np.maximum(s, f(0))
        i0 = s.astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (s - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, l1, (f(1) - l1).astype(np.float32)

    y0, y1, ly, ly0 = src(2 * h, h)
    x0, x1, lx, lx0 = src(2 * w, w)
    top = (lx0 * x[:, :, y0][:, :, :, x0]).astype(np.float32) + (lx * x[:, :, y0][:, :, :, x1]).astype(np.float32)
    bot = (lx0 * x[:, :, y1][:, :, :, x0]).astype(np.float32) + (lx * x[:, :, y1][:, :, :, x1]).astype(np.float32)
    up = ((ly0[:, None] * top).astype(np.float32) + (ly[:, None] * bot).astype(np.float32)).astype(np.float32)
    if C == 1:
        unc = -np.abs(up[:, 0])
    else:
        srt = np.sort(up, axis=1)
        unc = (srt[:, -2] - srt[:, -1]).astype(np.float32)
    return up, unc.reshape(N, -1).astype(np.float32)


def pr_topk(unc, k):
    """the k largest per row, exactly: ties at the k-th value go to the lowest indices; order = the strictly larger
    ones in index order, then the ties in index order (what emp_pr_topk emits; torch.topk's own tie choice and order
    are unspecified, point_rend.py:117 only uses the SET)"""
    unc = np.asarray(unc, dtype=np.float32)
    out = np.empty((unc.shape[0], k), dtype=np.int32)
    for n in range(unc.shape[0]):
        t = np.sort(unc[n])[-k]
        gt = np.flatnonzero(unc[n] > t)
        eq = np.flatnonzero(unc[n] == t)
        out[n] = np.concatenate([gt, eq[:k - len(gt)]])
    return out


def pr_point_sample(feat_nhwc, coarse, idx, H, W, ld):
    """point_sample (F.grid_sample bilinear, align_corners=False, zero padding; point_rend.py:35-60) of the features
    (N,Hf,Wf,CF) and the coarse logits (N,C,Hf,Wf) at the centres of the grid points idx (N,k) of an (H,W) grid ->
    X (N*k, ld) = [features | coarse | 0], the fp32 operations in emp_pr_point_sample's order."""
    feat = np.ascontiguousarray(feat_nhwc, dtype=np.float32)
    coarse = np.ascontiguousarray(coarse, dtype=np.float32)
    N, Hf, Wf, CF = feat.shape
    C = coarse.shape[1]
    k = idx.shape[1]
    f = np.float32
    px, py = (idx % W).astype(np.float32), (idx // W).astype(np.float32)
    cx = (f(0.5) / f(W) + px / f(W)).astype(np.float32)
    cy = (f(0.5) / f(H) + py / f(H)).astype(np.float32)
    gx, gy = (f(2) * cx - f(1)).astype(np.float32), (f(2) * cy - f(1)).astype(np.float32)
    ix = (((gx + f(1)) * f(Wf) - f(1)) / f(2)).astype(np.float32)
    iy = (((gy + f(1)) * f(Hf) - f(1)) / f(2)).astype(np.float32)
    x0, y0 = np.floor(ix).astype(np.int64), np.floor(iy).astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    wts = [((x1.astype(np.float32) - ix) * (y1.astype(np.float32) - iy)).astype(np.float32),
           ((ix - x0.astype(np.float32)) * (y1.astype(np.float32) - iy)).astype(np.float32),
           ((x1.astype(np.float32) - ix) * (iy - y0.astype(np.float32))).astype(np.float32),
           ((ix - x0.astype(np.float32)) * (iy - y0.astype(np.float32))).astype(np.float32)]
    nbr = [(y0, x0), (y0, x1), (y1, x0), (y1, x1)]
    X = np.zeros((N * k, ld), dtype=np.float32)
    n_of = np.repeat(np.arange(N), k)
    for (yy, xx), wt in zip(nbr, wts):
        ok = ((yy >= 0) & (yy < Hf) & (xx >= 0) & (xx < Wf)).ravel()
        yc, xc = np.clip(yy, 0, Hf - 1).ravel(), np.clip(xx, 0, Wf - 1).ravel()
        wv = np.where(ok, wt.ravel(), f(0)).astype(np.float32)
        # the kernel skips neighbours outside the map; adding 0 * v here changes nothing but the sign of a zero,
        # which a later addition of a non-zero term or the comparison as uint32 of +0 / -0 could expose: mask instead
        fv = (feat[n_of, yc, xc] * wv[:, None]).astype(np.float32)
        X[:, :CF] = np.where(ok[:, None], (X[:, :CF] + fv).astype(np.float32), X[:, :CF])
        cv = (coarse[n_of, :, yc, xc] * wv[:, None]).astype(np.float32)
        X[:, CF:CF + C] = np.where(ok[:, None], (X[:, CF:CF + C] + cv).astype(np.float32), X[:, CF:CF + C])
    return X
