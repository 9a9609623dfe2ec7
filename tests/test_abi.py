"""CPU: the C-ABI library loads and exports every symbol include/emp_hip.h declares; the product
package never imports the oracle."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'emp_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(emp_[a-z0-9_]+)\s*\(', text)))


def test_library_builds_and_exports_every_declared_symbol():
    from empanada_amd import build, _hip
    build.build(verbose=False)
    lib = _hip.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in emp_hip.h but not exported"
        assert name in _hip.SIGNATURES, f"{name} has no ctypes signature in empanada_amd/_hip.py"
    assert set(_hip.SIGNATURES) == set(declared)
    assert lib.emp_version() >= 100


def test_argument_validation_without_gpu():
    """error paths return before any launch, so they can be exercised on the CPU-only builder."""
    from empanada_amd import _hip
    lib = _hip.load()
    assert lib.emp_median_harden_stack(None, 4, 1, 16, 3, 0.5, None, None, None) == -1
    assert b'null' in lib.emp_last_error()
    assert lib.emp_find_centers(1, 1, 8, 8, 0.1, 99, 16, 1, 1, None) == -1
    assert lib.emp_group_pixels(1, 1, 16, 1, 1, 8, 8, 3, None, 0, 8, 1, None) == -1
    assert lib.emp_median_step(None, 4, 10, None, None) == -1


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from empanada_amd import _hip
    from empanada_amd.inference import rle
    with pytest.raises(_hip.HipError):
        rle.pan_seg_to_rle_seg(np.zeros((4, 4), np.int64), [1], 1000, [1])


def test_no_oracle_in_product():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'empanada_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f"{f} imports oracle"
                assert '/root/reference' not in src, f"{f} reads the reference at run time"
