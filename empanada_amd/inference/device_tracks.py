"""Device-resident trackers of one plane (the MI355X-native form of ``InstanceTracker``).

The reference accumulates, slice by slice, Python lists of numpy run arrays per instance
(``empanada/inference/tracker.py:61-123``) and hands three such trackers to the consensus.  At 1024^3 that is
~4 M runs per plane crossing PCIe and being sorted / split / concatenated by numpy on one core -- as long as a
third of the plane's forward pass.  Here a plane's trackers are two parts:

  * a small host table with one row per instance (label, class, 3D box, voxel count, position in the tracker's
    dict order) derived from the O(#components) tables the chain already works on, and
  * device arrays ``key`` (instance << 40 | flat start), ``st``, ``ln`` holding every 3D run of the plane sorted by
    (instance, start), produced from the run table by ``emp_track_lift`` / ``emp_track_lift_yz`` + ``emp_track_sort``.

The consensus and the fill read the device arrays directly (``empanada_amd/consensus.py``); ``PlaneTracks.trackers()``
materialises reference-ordered ``InstanceTracker`` objects for callers (and tests) that want the dicts.

With several ranks every rank lifts the runs of its own slices with GLOBAL instance indices (the chain is replicated,
so every rank holds the same instance table); ``gather_runs`` all-gathers them and ``clip_to_slab`` keeps the part
inside the rank's z-slab of the output volume.
"""
import numpy as np
import torch

from .. import _hip
from .tracker import InstanceTracker

__all__ = ['PlaneTracks', 'plane_tracks', 'POS_BITS']

POS_BITS = 40
_AXIS = {'xy': 0, 'xz': 1, 'yz': 2}


def _i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


class PlaneTracks:
    """Finished trackers of one plane, all classes: host instance table + device run arrays.

    Instance order = class order of ``labels``, then the tracker's dict order (order of first ``update`` when the
    slices are walked last to first, patterns.py:115-121).  ``alive`` is what the size / span filters leave."""

    def __init__(self, axis, shape3d, labels, label_divisor):
        self.axis = axis
        self.shape3d = tuple(int(s) for s in shape3d)
        self.labels = list(labels)
        self.label_divisor = int(label_divisor)
        self.inst_label = np.zeros(0, np.int64)
        self.inst_cls = np.zeros(0, np.int64)
        self.inst_area = np.zeros(0, np.int64)
        self.inst_box = np.zeros((0, 6), np.int64)
        self.alive = np.zeros(0, bool)
        self.inst_base = 0            # index of this plane's first instance among all planes (consensus objects)
        self.key = self.st = self.ln = None        # device: runs of THIS rank's slices, sorted by (instance, start)
        self.n_runs = 0

    @property
    def n_inst(self):
        return len(self.inst_label)

    def class_range(self, class_id):
        idx = np.flatnonzero(self.inst_cls == class_id)
        return (int(idx[0]), int(idx[-1]) + 1) if len(idx) else (0, 0)

    # -- filters (inference/filters.py:9-43) on the instance table
    def remove_small_objects(self, min_size=64):
        self.alive &= ~(self.inst_area < min_size)

    def remove_pancakes(self, min_span=4):
        b = self.inst_box
        self.alive &= ~((b[:, 3:] - b[:, :3]) < min_span).any(axis=1)

    def offsets(self):
        """CSR offsets (device int64, n_inst + 1) of the instances in the local run arrays"""
        top = self.inst_base + self.n_inst                       # keys carry inst_base + instance
        off = torch.empty((top + 1,), dtype=torch.int64, device=self.ln.device)
        _hip.call('emp_track_offsets', _hip._ptr(self.key) if self.n_runs else None, self.n_runs, top,
                  _hip._ptr(off), _hip.stream())
        return off[self.inst_base:]

    def trackers(self):
        """Reference-ordered InstanceTracker per class (tracker.py:102-121 result): D2H of the run arrays + the
        run order of the reference (xy / xz: slices descending, 2D start ascending; yz: voxel order).  Only valid on a
        rank that holds every run of the plane (one rank, or after gather_runs)."""
        Z, Y, X = self.shape3d
        key = self.key[:self.n_runs].cpu().numpy().view(np.uint64) if self.n_runs else np.zeros(0, np.uint64)
        inst = (key >> np.uint64(POS_BITS)).astype(np.int64) - self.inst_base
        st = (key & np.uint64((1 << POS_BITS) - 1)).astype(np.int64)
        ln = self.ln[:self.n_runs].cpu().numpy() if self.n_runs else np.zeros(0, np.int64)
        off = np.searchsorted(inst, np.arange(self.n_inst + 1))
        out = []
        for l in self.labels:
            tr = InstanceTracker(l, self.label_divisor, self.shape3d, self.axis)
            lo, hi = self.class_range(l)
            for i in range(lo, hi):
                if not self.alive[i]:
                    continue
                s, r = st[off[i]:off[i + 1]], ln[off[i]:off[i + 1]]
                if self.axis == 'xy':
                    o = np.lexsort((s, -(s // (Y * X))))
                elif self.axis == 'xz':
                    o = np.lexsort(((s // (Y * X)) * X + s % X, -((s % (Y * X)) // X)))
                else:
                    o = slice(None)
                tr.instances[int(self.inst_label[i])] = {'box': tuple(int(b) for b in self.inst_box[i]),
                                                         'starts': s[o], 'runs': r[o]}
            tr.finished = True
            out.append(tr)
        return out


def instance_table(host, final, first_seen, axis, labels):
    """Host half: one row per instance from the component tables of the whole axis.

    host: c_slice (global slice index), c_area, c_box, c_cls per component; final: final label per component
    (0 = none); first_seen {class: {label: position in the tracker's dict order}}.
    Returns (inst_label, inst_cls, inst_area, inst_box (n, 6), comp_inst (n_comp,) int32 with -1 = no instance)."""
    c_slice, c_area, c_box, c_cls = host['c_slice'], host['c_area'], host['c_box'].astype(np.int64), host['c_cls']
    comp_inst = np.full(len(final), -1, dtype=np.int32)
    lab_all, cls_all, area_all, box_all = [], [], [], []
    base = 0
    for l in labels:
        sel = np.flatnonzero((c_cls == l) & (final > 0))
        if len(sel) == 0:
            continue
        uniq, inv = np.unique(final[sel], return_inverse=True)
        rank = np.array([first_seen[l][int(u)] for u in uniq], dtype=np.int64)     # dict-order position of each label
        order = np.argsort(rank, kind='stable')
        pos = np.empty(len(uniq), dtype=np.int64)
        pos[order] = np.arange(len(uniq))
        idx = pos[inv]                                                             # instance (within class) per comp
        n = len(uniq)
        comp_inst[sel] = (base + idx).astype(np.int32)
        area = np.bincount(idx, weights=c_area[sel], minlength=n).astype(np.int64)
        big = np.iinfo(np.int64).max
        lo2 = np.full((n, 2), big, dtype=np.int64)
        hi2 = np.zeros((n, 2), dtype=np.int64)
        np.minimum.at(lo2, idx, c_box[sel, :2])
        np.maximum.at(hi2, idx, c_box[sel, 2:])
        s_lo = np.full(n, big, dtype=np.int64)
        s_hi = np.zeros(n, dtype=np.int64)
        np.minimum.at(s_lo, idx, c_slice[sel])
        np.maximum.at(s_hi, idx, c_slice[sel] + 1)
        # to_box3d + merge_boxes (tracker.py:11-23, 92-96): the slice index goes where the plane's normal is
        k = _AXIS[axis]
        lo3 = np.insert(lo2, k, s_lo, axis=1)
        hi3 = np.insert(hi2, k, s_hi, axis=1)
        lab_all.append(uniq[order])
        cls_all.append(np.full(n, l, dtype=np.int64))
        area_all.append(area)
        box_all.append(np.concatenate([lo3, hi3], axis=1))
        base += n
    if not lab_all:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 6), np.int64), comp_inst
    return (np.concatenate(lab_all), np.concatenate(cls_all), np.concatenate(area_all), np.concatenate(box_all),
            comp_inst)


def lift_runs(table, comp_inst_local, axis, shape3d, slice0=0, inst_base=0):
    """Device half: the rank's run table + instance index of each of its components -> (key, st, ln, n) sorted by
    (instance, start).  comp_inst_local: int32 numpy, -1 = skip (halo slice, no instance)."""
    _hip.require_gpu()
    Z, Y, X = (int(s) for s in shape3d)
    dev = table.r_start.device
    n_runs = int(table.n_runs)
    comp_inst = _i32(comp_inst_local if len(comp_inst_local) else np.zeros(1, np.int32), dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    st = _hip.stream()
    if axis in ('xy', 'xz'):
        cap = max(n_runs, 1)
        key = torch.empty((cap,), dtype=torch.int64, device=dev)
        ln = torch.empty((cap,), dtype=torch.int64, device=dev)
        work = torch.empty((_hip.query('emp_track_work_elems', n_runs),), dtype=torch.int32, device=dev)
        _hip.call('emp_track_lift', _AXIS[axis], _hip._ptr(table.r_start), _hip._ptr(table.r_len),
                  _hip._ptr(table.r_comp), _hip._ptr(table.c_slice), _hip._ptr(comp_inst), n_runs, table.H, table.W,
                  Y, X, int(slice0), int(inst_base), _hip._ptr(work), _hip._ptr(key), _hip._ptr(ln), _hip._ptr(cnt), st)
        n = int(cnt.item())
        merge = 0
    else:
        # scatter instance + 1 into a (Z, Y, Xl) volume, read its row runs back (the RLE along x of the dense
        # labelling), map them into the (Z, Y, X) frame
        Xl = table.D
        value = _hip.as_u32((comp_inst + 1).contiguous())
        vol = torch.zeros((Z, Y, Xl), dtype=torch.int32, device=dev).view(torch.uint32)
        _hip.call('emp_scatter_yz_u32', _hip._ptr(vol), Z, Y, Xl, _hip._ptr(table.r_start), _hip._ptr(table.r_len),
                  _hip._ptr(table.r_comp), _hip._ptr(table.c_slice), _hip._ptr(value), n_runs, st)
        rows = torch.empty((Z * Y,), dtype=torch.int32, device=dev)
        _hip.call('emp_runs_count', _hip._ptr(vol), Z, Y, Xl, _hip._ptr(rows), st)
        offs = _hip.exclusive_scan_i32(rows)
        n = int(offs[-1].item())
        cap = max(n, 1)
        r_start = torch.empty((cap,), dtype=torch.int32, device=dev)
        r_len = torch.empty_like(r_start)
        r_val = torch.empty((cap,), dtype=torch.uint32, device=dev)
        _hip.call('emp_runs_extract', _hip._ptr(vol), Z, Y, Xl, _hip._ptr(offs), _hip._ptr(r_start), _hip._ptr(r_len),
                  _hip._ptr(r_val), st)
        del vol
        key = torch.empty((cap,), dtype=torch.int64, device=dev)
        ln = torch.empty((cap,), dtype=torch.int64, device=dev)
        _hip.call('emp_track_lift_yz', _hip._ptr(offs), _hip._ptr(r_start), _hip._ptr(r_len), _hip._ptr(r_val), Z * Y,
                  n, Xl, X, int(slice0), int(inst_base), _hip._ptr(key), _hip._ptr(ln), st)
        merge = 1                                  # runs that touch across a row end are one run (np.sort + rle_encode)
    return sort_runs(key, ln, n, merge)


def sort_runs(key, ln, n, merge_touching=False):
    """(key, len)[:n] device arrays -> (key, st, ln, n') sorted by key; optionally joined where they touch"""
    dev = key.device
    cap = max(n, 1)
    okey = torch.empty((cap,), dtype=torch.int64, device=dev)
    ost = torch.empty((cap,), dtype=torch.int64, device=dev)
    oln = torch.empty((cap,), dtype=torch.int64, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    wb = _hip.query('emp_track_sort_work_bytes', n)
    work = torch.empty((wb,), dtype=torch.uint8, device=dev)
    _hip.call('emp_track_sort', _hip._ptr(key), _hip._ptr(ln), n, int(bool(merge_touching)), _hip._ptr(work), wb,
              _hip._ptr(okey), _hip._ptr(ost), _hip._ptr(oln), _hip._ptr(cnt), _hip.stream())
    m = int(cnt.item()) if merge_touching else n
    return okey[:m], ost[:m], oln[:m], m


def clip_runs(key, ln, n, lo, hi):
    """part of every run inside the flat voxel interval [lo, hi) -> (key, ln, n'), order preserved"""
    dev = key.device
    cap = max(n, 1)
    okey = torch.empty((cap,), dtype=torch.int64, device=dev)
    oln = torch.empty((cap,), dtype=torch.int64, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    work = torch.empty((_hip.query('emp_track_work_elems', n),), dtype=torch.int32, device=dev)
    _hip.call('emp_track_clip', _hip._ptr(key), _hip._ptr(ln), n, int(lo), int(hi), _hip._ptr(work), _hip._ptr(okey),
              _hip._ptr(oln), _hip._ptr(cnt), _hip.stream())
    m = int(cnt.item())
    return okey[:m], oln[:m], m


def plane_tracks(table, host, final, first_seen, axis, shape3d, labels, label_divisor, slice0=0, inst_base=0,
                 local_comp_index=None):
    """Build the PlaneTracks of one plane.

    host / final / first_seen describe the WHOLE axis (every rank holds them: the chain is replicated);
    table is the rank's device run table and local_comp_index[i] the row of `host` that is the table's component i
    (-1 for halo components); None = the table covers the whole axis in the same component order."""
    pt = PlaneTracks(axis, shape3d, labels, label_divisor)
    pt.inst_label, pt.inst_cls, pt.inst_area, pt.inst_box, comp_inst = instance_table(host, final, first_seen, axis,
                                                                                      labels)
    pt.alive = np.ones(pt.n_inst, dtype=bool)
    pt.inst_base = int(inst_base)
    if local_comp_index is not None:
        idx = np.asarray(local_comp_index)
        local = np.full(len(idx), -1, dtype=np.int32)
        ok = idx >= 0
        local[ok] = comp_inst[idx[ok]]
        comp_inst = local
    pt.key, pt.st, pt.ln, pt.n_runs = lift_runs(table, comp_inst, axis, shape3d, slice0, inst_base)
    return pt
