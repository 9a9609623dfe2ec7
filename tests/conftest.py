import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def dense_tol(ref_absmax):
    """D1 tolerance (fp32 logits rtol 1e-4): max |got - ref| <= 1e-4 * |ref|_inf + 1e-6 per head tensor.
    Relative to the tensor's own full scale -- the synthetic heads put out |logits|_inf of 0.03-0.07, so a floor like
    max(1, |ref|_inf) would accept errors of 0.3 % of full scale."""
    return 1e-4 * float(ref_absmax) + 1e-6


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def unpack_rle_seg(g, prefix, with_cls=True):
    """inverse of oracle/gen_golden.py:_pack_rle_seg -> {class: {label: attrs}} (dict order kept)."""
    lab, box, off = g[f'{prefix}_lab'], g[f'{prefix}_box'], g[f'{prefix}_off']
    starts, runs = g[f'{prefix}_starts'], g[f'{prefix}_runs']
    cls = g[f'{prefix}_cls'] if with_cls else np.zeros(len(lab), dtype=np.int64)
    out = {}
    for i in range(len(lab)):
        out.setdefault(int(cls[i]), {})[int(lab[i])] = {
            'box': tuple(int(b) for b in box[i]),
            'starts': starts[off[i]:off[i + 1]], 'runs': runs[off[i]:off[i + 1]]}
    return out


def unpack_instances(g, prefix):
    d = unpack_rle_seg(g, prefix, with_cls=False)
    return d.get(0, {})


def assert_instances_equal(a, b, check_order=True):
    assert list(a.keys()) == list(b.keys()) if check_order else set(a) == set(b)
    for k in a:
        assert tuple(int(x) for x in a[k]['box']) == tuple(int(x) for x in b[k]['box']), k
        np.testing.assert_array_equal(np.asarray(a[k]['starts']), np.asarray(b[k]['starts']), err_msg=str(k))
        np.testing.assert_array_equal(np.asarray(a[k]['runs']), np.asarray(b[k]['runs']), err_msg=str(k))


@pytest.fixture(scope='session')
def golden():
    return load_golden
