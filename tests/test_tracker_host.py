"""T1 / T2 host logic that needs no GPU: the product InstanceTracker on xy and xz slices (yz decodes pixels through the
HIP run decoder and is covered by the GPU tests) against the oracle's tracker, and the JSON wire format
(tracker.py:125-159: key order of the dumped attribute dict, "s r s r" run strings, string instance keys on load)."""
import json

import numpy as np
import pytest

from empanada_amd.inference.tracker import InstanceTracker, to_box3d, to_coords3d
from oracle import rle_seg as OS


def _slices(rng, n_slices, plane_shape, labels):
    out = []
    size = plane_shape[0] * plane_shape[1]
    for _ in range(n_slices):
        seg = {}
        for lab in labels:
            if rng.random() < 0.3:
                continue
            n = int(rng.integers(1, 6))
            starts = np.sort(rng.choice(size - 8, n, replace=False)).astype(np.int64)
            runs = rng.integers(1, 8, n).astype(np.int64)
            r0, c0 = (int(v) for v in rng.integers(0, 5, 2))
            seg[lab] = {'box': (r0, c0, r0 + int(rng.integers(1, 5)), c0 + int(rng.integers(1, 5))),
                        'starts': starts, 'runs': runs}
        out.append(seg)
    return out


@pytest.mark.parametrize('axis', ['xy', 'xz'])
def test_tracker_matches_oracle(axis):
    rng = np.random.default_rng(5)
    shape3d = (7, 9, 11)
    k = {'xy': 0, 'xz': 1}[axis]
    plane = tuple(s for i, s in enumerate(shape3d) if i != k)
    segs = _slices(rng, shape3d[k], plane, [20001, 20002, 20007])
    got, exp = InstanceTracker(1, 20000, shape3d, axis), OS.InstanceTracker(1, 20000, shape3d, axis)
    for idx in reversed(range(len(segs))):                     # the backward pass walks last to first
        got.update(segs[idx], idx)
        exp.update(segs[idx], idx)
    got.finish()
    exp.finish()
    assert list(got.instances) == list(exp.instances)
    for lab in exp.instances:
        assert tuple(got.instances[lab]['box']) == tuple(exp.instances[lab]['box'])
        for key in ('starts', 'runs'):
            assert got.instances[lab][key].dtype == np.int64
            np.testing.assert_array_equal(got.instances[lab][key], exp.instances[lab][key])
    with pytest.raises(AssertionError):
        got.update(segs[0], 0)                                 # finished


def test_box_and_coords_lifting():
    assert to_box3d(4, (1, 2, 3, 5), 'xy') == (4, 1, 2, 5, 3, 5)
    assert to_box3d(4, (1, 2, 3, 5), 'xz') == (1, 4, 2, 3, 5, 5)
    assert to_box3d(4, (1, 2, 3, 5), 'yz') == (1, 2, 4, 3, 5, 5)
    r, c = np.array([0, 1]), np.array([5, 6])
    for axis, order in (('xy', (2, 0, 1)), ('xz', (0, 2, 1)), ('yz', (0, 1, 2))):
        parts = (r, c, np.array([9, 9]))
        got = to_coords3d(9, (r, c), axis)
        for g, j in zip(got, order):
            np.testing.assert_array_equal(g, parts[j])
    with pytest.raises(AssertionError):
        to_box3d(0, (0, 0, 1, 1), 'zz')


def test_json_wire_format(tmp_path):
    tr = InstanceTracker(2, 1000, (4, 5, 6), 'xy')
    tr.update({2001: {'box': (0, 1, 2, 3), 'starts': np.array([3, 10]), 'runs': np.array([2, 4])}}, 1)
    tr.update({2001: {'box': (1, 0, 3, 2), 'starts': np.array([7]), 'runs': np.array([1])},
               2005: {'box': (0, 0, 1, 1), 'starts': np.array([0]), 'runs': np.array([1])}}, 0)
    path = tmp_path / 't.json'
    tr.write_to_json(str(path))                                # finishes the tracker first
    text = path.read_text()
    doc = json.loads(text)
    assert list(doc) == ['class_id', 'label_divisor', 'shape3d', 'axis', 'finished', 'instances', 'axis_nums']
    assert doc['finished'] is True and doc['axis_nums'] == {'xy': 0, 'xz': 1, 'yz': 2}
    assert list(doc['instances']) == ['2001', '2005'] and list(doc['instances']['2001']) == ['box', 'rle']
    assert doc['instances']['2001'] == {'box': [0, 0, 0, 2, 3, 3], 'rle': '33 2 40 4 7 1'}
    assert text.startswith('{\n      "class_id": 2,')          # indent 6
    # the tracker itself is untouched by the dump, and the file loads back with string keys (as in the reference)
    np.testing.assert_array_equal(tr.instances[2001]['starts'], [33, 40, 7])
    back = InstanceTracker()
    back.load_from_json(str(path))
    assert back.class_id == 2 and back.axis == 'xy' and back.finished and back.shape3d == [4, 5, 6]
    np.testing.assert_array_equal(back.instances['2001']['starts'], [33, 40, 7])
    np.testing.assert_array_equal(back.instances['2001']['runs'], [2, 4, 1])
