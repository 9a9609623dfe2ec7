"""3D instance tracker, reference names and semantics (``empanada/inference/tracker.py``):
``InstanceTracker`` :40-159 (update :61-100, finish :102-123, JSON :125-159), ``to_box3d`` :11-23.

The tracker is an O(#runs) accumulator of index arrays; like in the reference it is host-side
bookkeeping (numpy).  The xz row-wrap behaviour of tracker.py:78-82 (only run *starts* are mapped to
3D) is reproduced on purpose: results must be bit-identical to the reference.
"""
import json
import math
from copy import deepcopy

import numpy as np

from ..array_utils import merge_boxes, rle_decode, rle_encode, rle_to_string, string_to_rle

__all__ = ['InstanceTracker', 'to_box3d', 'to_coords3d']


def to_box3d(index2d, box, axis):
    """tracker.py:11-23"""
    assert axis in ['xy', 'xz', 'yz']
    h1, w1, h2, w2 = box
    if axis == 'xy':
        return (index2d, h1, w1, index2d + 1, h2, w2)
    if axis == 'xz':
        return (h1, index2d, w1, h2, index2d + 1, w2)
    return (h1, w1, index2d, h2, w2, index2d + 1)


def to_coords3d(index2d, coords, axis):
    """tracker.py:25-38"""
    assert axis in ['xy', 'xz', 'yz']
    hcoords, wcoords = coords
    dcoords = np.repeat([index2d], len(hcoords))
    if axis == 'xy':
        return (dcoords, hcoords, wcoords)
    if axis == 'xz':
        return (hcoords, dcoords, wcoords)
    return (hcoords, wcoords, dcoords)


class InstanceTracker:
    """tracker.py:40-159"""

    def __init__(self, class_id=None, label_divisor=None, shape3d=None, axis='xy'):
        assert axis in ['xy', 'xz', 'yz']
        self.class_id = class_id
        self.label_divisor = label_divisor
        self.shape3d = shape3d
        self.axis = axis
        self.finished = False
        self.reset()
        self.axis_nums = {'xy': 0, 'xz': 1, 'yz': 2}

    def reset(self):
        self.instances = {}

    def update(self, instance_rles, index2d):
        assert self.class_id is not None
        assert self.label_divisor is not None
        assert self.shape3d is not None
        assert not self.finished, "Cannot update tracker after calling finish!"
        ignore_idx = self.axis_nums[self.axis]
        shape2d = tuple(s for i, s in enumerate(self.shape3d) if i != ignore_idx)
        for label, attrs in instance_rles.items():
            box = to_box3d(index2d, attrs['box'], self.axis)
            if self.axis == 'xy':
                starts = attrs['starts'] + index2d * math.prod(shape2d)
                runs = attrs['runs']
            elif self.axis == 'xz':
                coords2d = np.unravel_index(attrs['starts'], shape2d)
                starts = np.ravel_multi_index(to_coords3d(index2d, coords2d, 'xz'), self.shape3d)
                runs = attrs['runs']
            else:
                coords2d = np.unravel_index(rle_decode(attrs['starts'], attrs['runs']), shape2d)
                starts = np.ravel_multi_index(to_coords3d(index2d, coords2d, 'yz'), self.shape3d)
                runs = np.ones_like(starts)
            if label not in self.instances:
                self.instances[label] = {'box': box, 'starts': [starts], 'runs': [runs]}
            else:
                inst = self.instances[label]
                inst['box'] = merge_boxes(box, inst['box'])
                inst['starts'].append(starts)
                inst['runs'].append(runs)

    def finish(self):
        for instance_id in self.instances.keys():
            inst = self.instances[instance_id]
            if isinstance(inst['starts'], list):
                starts = np.concatenate(inst['starts'])
                if self.axis == 'yz':
                    starts, runs = rle_encode(np.sort(starts, kind='stable'))
                else:
                    runs = np.concatenate(inst['runs'])
                inst['starts'] = starts
                inst['runs'] = runs
        self.finished = True

    def write_to_json(self, savepath):
        """tracker.py:125-146 (wire format: class_id, label_divisor, shape3d, axis, finished, axis_nums,
        instances{str(id): {box, rle "s r s r ..."}})."""
        if not self.finished:
            self.finish()
        save_dict = deepcopy(self.__dict__)
        for k in save_dict['instances'].keys():
            inst = save_dict['instances'][k]
            inst['rle'] = rle_to_string(inst['starts'], inst['runs'])
            del inst['starts']
            del inst['runs']
        for k, v in list(save_dict['instances'].items()):
            save_dict['instances'][str(k)] = v
            del save_dict['instances'][k]
        with open(savepath, mode='w') as handle:
            json.dump(save_dict, handle, indent=6)

    def load_from_json(self, fpath):
        """tracker.py:148-159"""
        with open(fpath, mode='r') as handle:
            load_dict = json.load(handle)
        for k in load_dict['instances'].keys():
            starts, runs = string_to_rle(load_dict['instances'][k]['rle'])
            load_dict['instances'][k]['starts'] = starts
            load_dict['instances'][k]['runs'] = runs
        self.__dict__ = load_dict
