"""3D instance tracker with the reference's names and semantics (``empanada/inference/tracker.py``:
``InstanceTracker`` :40-159, ``to_box3d`` :11-23, ``to_coords3d`` :25-38).

A tracker accumulates, per instance label, the 2D run lists of successive slices of one plane as flat indices
into the (Z, Y, X) volume.  It is O(#runs) host bookkeeping; the whole-stack path builds the same structures from
device tables (``inference/patterns.py``) and only uses this class as the container.

Index arithmetic (``shape3d`` = (Z, Y, X), slice index ``i`` along the plane's normal):

  xy  slice (Y, X):  flat3d = flat2d + i * Y * X                        runs unchanged
  xz  slice (Z, X):  flat3d = (flat2d // X) * Y * X + i * X + flat2d % X  runs unchanged -- only the run STARTS
      are mapped, so a run that wraps over the end of a 2D row continues along x in 3D instead of jumping to the
      next z; this is what the reference does (tracker.py:78-82) and results must be bit-identical to it
  yz  slice (Z, Y):  every pixel p of every run: flat3d = (p // Y) * Y * X + (p % Y) * X + i, runs of length 1;
      ``finish`` sorts them and re-encodes (tracker.py:83-90, 108-117)

Slices arrive in whatever order the caller walks them (the backward matching pass walks last to first), and the
per-instance pieces are concatenated in arrival order (tracker.py:92-100).
"""
import json

import numpy as np

from ..array_utils import merge_boxes, rle_decode, rle_encode, rle_to_string, string_to_rle
from .deferred import LazyClass, LazyInstances

__all__ = ['InstanceTracker', 'to_box3d', 'to_coords3d']

_NORMAL = {'xy': 0, 'xz': 1, 'yz': 2}          # position of the slice index among (z, y, x)


def _normal_of(axis):
    if axis not in _NORMAL:
        raise AssertionError(f"axis must be one of {sorted(_NORMAL)}, got {axis!r}")
    return _NORMAL[axis]


def to_box3d(index2d, box, axis):
    """(r0, c0, r1, c1) of a slice -> half-open (z0, y0, x0, z1, y1, x1); the slice is one voxel thick."""
    k = _normal_of(axis)
    lo, hi = [box[0], box[1]], [box[2], box[3]]
    lo.insert(k, index2d)
    hi.insert(k, index2d + 1)
    return tuple(lo + hi)


def to_coords3d(index2d, coords, axis):
    """(rows, cols) of a slice -> (z, y, x) coordinate arrays."""
    k = _normal_of(axis)
    out = [coords[0], coords[1]]
    out.insert(k, np.full(len(coords[0]), index2d))
    return tuple(out)


class InstanceTracker:
    """Container + accumulator.  Attributes are part of the interface (the reference's callers assign
    ``instances`` directly and read ``class_id``, ``shape3d``, ``label_divisor``, ``axis``, ``finished``)."""

    def __init__(self, class_id=None, label_divisor=None, shape3d=None, axis='xy'):
        _normal_of(axis)
        # attribute order = key order of the JSON file (tracker.py:46-55 dumps __dict__)
        self.class_id = class_id
        self.label_divisor = label_divisor
        self.shape3d = shape3d
        self.axis = axis
        self.finished = False
        self.instances = {}
        self.axis_nums = dict(_NORMAL)

    def reset(self):
        self.instances = {}

    # ------------------------------------------------------------------ accumulation
    def _lift(self, starts, runs, index2d):
        """2D (starts, runs) of slice index2d -> 3D (starts, runs), int64, per the table in the module docstring"""
        _, Y, X = (int(s) for s in self.shape3d)
        starts = np.asarray(starts)
        if self.axis == 'xy':
            return starts + index2d * (Y * X), runs
        if self.axis == 'xz':
            s = starts.astype(np.int64, copy=False)
            return (s // X) * (Y * X) + index2d * X + s % X, runs
        p = np.asarray(rle_decode(starts, runs)).astype(np.int64, copy=False)
        lifted = (p // Y) * (Y * X) + (p % Y) * X + index2d
        return lifted, np.ones_like(lifted)

    def update(self, instance_rles, index2d):
        """instance_rles: {label: {'box', 'starts', 'runs'}} of one slice (inference/rle.py), index2d its position."""
        missing = [n for n in ('class_id', 'label_divisor', 'shape3d') if getattr(self, n) is None]
        assert not missing, f"tracker is missing {missing}"
        assert not self.finished, "Cannot update tracker after calling finish!"
        if isinstance(instance_rles, LazyClass) and instance_rles._s.bwd == 'lazy':
            # a handle of a deferred stack (inference/deferred.py): filed by finish(), or by the first look inside
            held = self.instances
            if type(held) is dict and not held:
                held = self.instances = LazyInstances(self)
            if isinstance(held, LazyInstances) and not held._resolved:
                held.record(instance_rles, index2d)
                return
        self._update_now(instance_rles, index2d)

    def _update_now(self, instance_rles, index2d):
        for label, piece in instance_rles.items():
            box = to_box3d(index2d, piece['box'], self.axis)
            starts, runs = self._lift(piece['starts'], piece['runs'], index2d)
            entry = self.instances.get(label)
            if entry is None:
                self.instances[label] = {'box': box, 'starts': [starts], 'runs': [runs]}
                continue
            entry['box'] = merge_boxes(box, entry['box'])
            entry['starts'].append(starts)
            entry['runs'].append(runs)

    def finish(self):
        """lists of per-slice pieces -> one (starts, runs) pair per instance; yz pixels are sorted and re-encoded"""
        held = self.instances
        if isinstance(held, LazyInstances):
            session = held.fast_ok()
            if session is not None:
                # every slice of a deferred stack, last to first, and nobody looked: the whole-stack path
                final = dict(session.tracker_instances(self.axis, self.shape3d, self.class_id))
                held.fill(final)
                self.instances = final
                self.finished = True
                return
            held.resolve()
            self.instances = dict.copy(held)
        for entry in self.instances.values():
            if not isinstance(entry['starts'], list):
                continue                                   # assigned in final form by a caller
            flat = np.concatenate(entry['starts'])
            if self.axis == 'yz':
                entry['starts'], entry['runs'] = rle_encode(np.sort(flat, kind='stable'))
            else:
                entry['starts'], entry['runs'] = flat, np.concatenate(entry['runs'])
        self.finished = True

    # ------------------------------------------------------------------ wire format (tracker.py:125-159)
    def write_to_json(self, savepath):
        """{class_id, label_divisor, shape3d, axis, finished, instances: {"<id>": {box, rle: "s r s r ..."}},
        axis_nums}, indent 6 -- byte-identical to the reference's file for equal content."""
        if not self.finished:
            self.finish()
        doc = {}
        for name, value in self.__dict__.items():
            if name != 'instances':
                doc[name] = value
                continue
            doc[name] = {}
            for label, entry in value.items():
                packed = {k: v for k, v in entry.items() if k not in ('starts', 'runs')}
                packed['rle'] = rle_to_string(entry['starts'], entry['runs'])
                doc[name][str(label)] = packed
        with open(savepath, mode='w') as fh:
            json.dump(doc, fh, indent=6)

    def load_from_json(self, fpath):
        """Inverse of write_to_json.  As in the reference the loaded document BECOMES the attribute dict: instance
        keys stay strings and boxes / shape3d come back as lists."""
        with open(fpath, mode='r') as fh:
            doc = json.load(fh)
        for entry in doc['instances'].values():
            entry['starts'], entry['runs'] = string_to_rle(entry['rle'])
        self.__dict__ = doc
