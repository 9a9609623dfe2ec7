"""Loader/builder for the plain-C half of the oracle (oracle/c/oracle_kernels.c).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "c", "oracle_kernels.c")
_SO = os.path.join(_HERE, "c", "liboracle_kernels.so")

_lib = None


def build(force=False):
    """Compile the C oracle with gcc (no fp contraction; explicit fmaf only)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
               "-o", _SO, _SRC, "-lm"]
        subprocess.run(cmd, check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64p = ctypes.POINTER(ctypes.c_int64)
        f32p = ctypes.POINTER(ctypes.c_float)
        L.emp_oracle_find_centers.restype = ctypes.c_int64
        L.emp_oracle_find_centers.argtypes = [f32p, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                              ctypes.c_int, i64p, ctypes.c_int64]
        L.emp_oracle_group_pixels.restype = None
        L.emp_oracle_group_pixels.argtypes = [i64p, ctypes.c_int64, f32p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, i64p]
        L.emp_oracle_cc8.restype = ctypes.c_int64
        L.emp_oracle_cc8.argtypes = [i64p, ctypes.c_int, ctypes.c_int, i64p]
        L.emp_oracle_median.restype = None
        L.emp_oracle_median.argtypes = [ctypes.POINTER(f32p), ctypes.c_int, ctypes.c_int64, f32p]
        L.emp_oracle_fill_u32.restype = None
        L.emp_oracle_fill_u32.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int64, i64p,
                                          i64p, ctypes.c_int64, ctypes.c_uint32]
        L.emp_oracle_dwconv_nhwc.restype = None
        L.emp_oracle_dwconv_nhwc.argtypes = [f32p, f32p, f32p] + [ctypes.c_int] * 5 + [f32p]
        L.emp_oracle_conv_bn_act_nhwc.restype = None
        L.emp_oracle_conv_bn_act_nhwc.argtypes = [f32p] * 5 + [ctypes.c_int] * 12 + [f32p]
        L.emp_oracle_conv_splitk_bn_act_nhwc.restype = None
        L.emp_oracle_conv_splitk_bn_act_nhwc.argtypes = [f32p] * 5 + [ctypes.c_int] * 12 + [f32p]
        L.emp_oracle_conv7s2_c1.restype = None
        L.emp_oracle_conv7s2_c1.argtypes = [f32p, f32p] + [ctypes.c_int] * 4 + [f32p]
        L.emp_oracle_gconv3x3_bn_act_nhwc.restype = None
        L.emp_oracle_gconv3x3_bn_act_nhwc.argtypes = [f32p] * 4 + [ctypes.c_int] * 8 + [f32p]
        _lib = L
    return _lib
