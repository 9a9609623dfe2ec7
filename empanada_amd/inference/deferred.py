"""Deferred evaluation of the per-slice (drop-in) protocol: the calls of scripts/pdl_inference3d.py:163-198 --
``engine(image)`` -> ``pan_seg_to_rle_seg`` -> ``apply_matchers`` -> ``engine.end()`` -> ``backward_matching`` ->
``update_trackers`` -> ``finish_tracking`` -- hand one slice from function to function, and every function of the
reference computes on the spot.  On this hardware that costs a batch-1 forward on 256 CUs plus a host round trip per
slice and per stage (tools/bench_per_slice.py: 5.9 ms per 512^2 slice against 0.9 ms for the whole-stack path).

With ``deferred=True`` a 3d engine hands out HANDLES instead of values: ``LazyPan`` for the panoptic image of a slice,
``LazySeg`` for its run-length segmentation (before and after forward matching), ``LazyFinal`` / ``LazyClass`` for the
backward-matched segmentation that ``update_trackers`` files, and the trackers collect handles too.  The functions
above recognise the handles and only RECORD what was asked.  The work is done once, when ``finish_tracking`` (or
anything that looks inside a handle) needs a value:

  * nobody looked (the standard call sequence): the images went through the model in batches, the heads of the whole
    stack sit in HBM, and the whole-stack path (``postprocess.panoptic_stack`` -> ``patterns.tables_from_stack`` ->
    ``chain_from_tables`` -> ``device_tracks.plane_tracks``) fills the trackers -- the same path bench.py times and
    tests/test_pipeline_gpu.py pins to the per-slice protocol;
  * somebody looked (``np.asarray(pan)``, ``seg[1]``, pickling for an mp.Queue, an unexpected call order, arguments that
    change from slice to slice): the recorded calls are replayed in order through the per-slice functions themselves,
    on the heads already computed, up to the point that is needed; the matchers, the median queue and the trackers end
    up in the state the reference leaves them in, and from then on everything is computed on the spot.

So the values are those of the per-slice protocol whichever way they are obtained (up to the float boundary of
DESIGN.md: the forward runs in batches of ``deferred_batch`` images rather than one).
"""
import functools
import threading

import numpy as np
import torch

__all__ = ['StackSession', 'LazyPan', 'LazySeg', 'LazyFinal', 'LazyClass', 'LazyInstances']

_HEADS = ('sem', 'ctr_hmp', 'offsets')
_PAN_OPS = ('squeeze', 'cpu', 'numpy', 'detach', 'contiguous')


class _Pending:
    """what a matcher's ``target_rle`` holds while its slices are deferred (not None: the matcher is initialised).
    Pickling the matcher (it may cross an mp.Queue) files the recorded calls first and stores the real target."""

    def __init__(self, session, stage, owner=None):
        self.session, self.stage, self.owner = session, stage, owner

    def resolve(self):
        """file the recorded calls: afterwards the owning matcher holds its real target and label counter"""
        if self.stage == 'fwd':
            self.session._forward_now()
        else:
            self.session._backward_now()

    def __reduce__(self):
        self.resolve()
        real = self.owner.target_rle if self.owner is not None else None
        return (_same, (None if isinstance(real, _Pending) else real,))

    def _die(self, *a, **k):
        raise RuntimeError("this matcher's target belongs to a deferred stack; use the matcher through "
                           "apply_matchers / backward_matching or build the engine with deferred=False")

    keys = items = values = __iter__ = __len__ = __getitem__ = _die


def _same(x):
    return x


def _locked(method):
    """session state is touched from wherever a handle is looked at -- an mp.Queue pickles in its feeder THREAD"""
    @functools.wraps(method)
    def guarded(self, *args, **kwargs):
        with self._lock:
            return method(self, *args, **kwargs)
    return guarded


class StackSession:
    """One stack of slices handed to a 3d engine one image at a time."""

    def __init__(self, engine, batch_size=16):
        self._lock = threading.RLock()
        self.engine = engine
        self.batch = max(int(batch_size), 1)
        self.ks, self.m = engine.ks, engine.mid_idx
        self.calls = 0
        self.closed = False
        self.n_emitted = 0
        self.images, self.sizes = [], []
        self.size = self.upsampling = self.shape = None
        self.uniform = True
        self.n_fwd = 0
        self.chunks = []                       # (first call, end call, {head: (n, ...) fp32 tensor})
        self.chunk_of = []                     # call -> index into chunks
        # replay cursor of the per-slice engine code (uses the engine's own median queue)
        self.eager_fed = 0
        self.eager_out = []
        self.eager_ended = False
        # rle / forward matching stage
        self.rle_args = None
        self.segs = {}
        self.matchers = None
        self.matcher_init = None
        self.n_matched = 0
        self.fwd_lazy = True
        self.eager_matched = 0
        # backward stage
        self.bwd = None                        # None | 'lazy' | 'eager'
        self.bwd_real = {}
        # whole-stack results
        self._pan = self._tables = self._chain = None
        self._tracks = {}

    # ------------------------------------------------------------------ engine side
    @_locked
    def add(self, image, size=None, upsampling=1):
        assert not self.closed
        if self.calls == 0:
            self.shape, self.upsampling = tuple(image.shape), upsampling
            self.size = None if size is None else (int(size[0]), int(size[1]))
        elif self.uniform:
            same = tuple(image.shape) == self.shape and upsampling == self.upsampling and (
                (size is None) == (self.size is None)) and (size is None or (int(size[0]), int(size[1])) == self.size)
            if not same:
                self.uniform = False           # slices of different shapes: no stack to work on, values on demand
        self.images.append(image)
        self.sizes.append(size)
        self.calls += 1
        if self.uniform and self.calls - self.n_fwd >= self.batch:
            self._flush(self.calls)
        held = min(self.calls, self.ks)
        if held == self.ks or held <= self.m:              # _MedianQueue.get_next: which calls hand a slice out
            return self._emit()
        return None

    def _emit(self):
        k, self.n_emitted = self.n_emitted, self.n_emitted + 1
        return LazyPan(self, k)

    @_locked
    def end(self):
        """_MedianQueue.end: the items right of the middle (none while the queue holds at most ks // 2 + 1)"""
        assert not self.closed
        self.closed = True
        return [self._emit() for _ in range(max(min(self.calls, self.ks) - 1 - self.m, 0))]

    @torch.no_grad()
    def _flush(self, upto):
        while self.n_fwd < upto:
            hi = min(self.n_fwd + self.batch, upto) if self.uniform else self.n_fwd + 1
            imgs = self.images[self.n_fwd:hi]
            x = imgs[0] if len(imgs) == 1 else torch.cat(imgs, dim=0)
            heads = self.engine._deferred_infer(x, self.upsampling)
            short = [k for k in _HEADS if heads[k].size(0) != hi - self.n_fwd]
            if short:
                raise RuntimeError(f"deferred engine: the model returned {heads[short[0]].size(0)} slices of "
                                   f"'{short[0]}' for a batch of {hi - self.n_fwd} images -- a model that answers one "
                                   f"image per call needs deferred_batch=1 (or deferred=False)")
            self.chunk_of += [len(self.chunks)] * (hi - self.n_fwd)
            self.chunks.append((self.n_fwd, hi, {k: heads[k].float() for k in _HEADS}))
            for i in range(self.n_fwd, hi):
                self.images[i] = None
            self.n_fwd = hi

    def _head_slice(self, t):
        lo, _, heads = self.chunks[self.chunk_of[t]]
        out = {k: heads[k][t - lo:t - lo + 1] for k in _HEADS}
        if self.sizes[t] is not None:
            out['size'] = self.sizes[t]
        return out

    # ------------------------------------------------------------------ values on demand (replay of the per-slice code)
    @_locked
    def force_pan(self, k):
        eng = self.engine
        while len(self.eager_out) <= k:
            if self.eager_fed < self.calls:
                self._flush(self.calls)
                eng.enqueue(self._head_slice(self.eager_fed))
                self.eager_fed += 1
                ready = eng.get_next(keys=['sem'])
                if ready is not None:
                    self.eager_out.append(eng._labels_now(ready, self.upsampling))
            elif self.closed and not self.eager_ended:
                self.eager_ended = True
                self.eager_out += eng._end_now(self.upsampling)
            else:
                raise RuntimeError("deferred engine: slice requested before the engine has seen enough images")
        return self.eager_out[k]

    @_locked
    def go_eager(self):
        """bring the engine's own median queue to where the reference's would be after the calls so far"""
        if self.n_emitted:
            self.force_pan(self.n_emitted - 1)
        while self.eager_fed < self.calls:                 # calls that handed nothing out still fill the queue
            self._flush(self.calls)
            self.engine.enqueue(self._head_slice(self.eager_fed))
            self.eager_fed += 1
            assert self.engine.get_next(keys=['sem']) is None

    def _rle_now(self, k):
        from . import rle
        seg = self.segs[k]
        if seg._real is None:
            pan = self.force_pan(k).squeeze().cpu().numpy()
            seg._real = rle.pan_seg_to_rle_seg(pan, *self.rle_args)
        return seg._real

    @_locked
    def _forward_now(self):
        """replay apply_matchers over every slice that went through it, in order, on the matchers themselves"""
        from . import patterns
        if self.fwd_lazy:
            self.fwd_lazy = False
            self.eager_matched = 0
            if self.matchers is not None:
                for mt, (nxt, new) in zip(self.matchers, self.matcher_init):
                    mt.target_rle, mt.next_label, mt.assign_new = None, nxt, new
        while self.eager_matched < self.n_matched:
            patterns._apply_matchers_now(self._rle_now(self.eager_matched), self.matchers)
            self.eager_matched += 1

    @_locked
    def force_seg(self, k):
        real = self._rle_now(k)
        if self.segs[k]._matched:
            self._forward_now()
        return real

    @_locked
    def _backward_now(self):
        from . import patterns
        if self.bwd == 'lazy':
            self.bwd = 'eager'
            self._forward_now()
            stack = [self._rle_now(k) for k in range(self.n_emitted)]
            for idx, rs in patterns._backward_matching_now(stack, self.matchers, self.n_emitted):
                self.bwd_real[idx] = rs

    @_locked
    def force_final(self, k):
        self._backward_now()
        return self.bwd_real[k]

    # ------------------------------------------------------------------ recording
    def _stack_ok(self):
        return self.uniform and self.fwd_lazy

    @_locked
    def lazy_rle(self, pan, args):
        """pan_seg_to_rle_seg on a handle -> LazySeg, or None when the call has to be computed now"""
        labels, div, things, force_connected = args
        if not (self.uniform and pan._is2d() and force_connected and pan._k not in self.segs):
            return None
        args = (list(labels), int(div), list(things), True)
        if self.rle_args is None:
            self.rle_args = args
        elif self.rle_args != args:
            return None
        seg = self.segs[pan._k] = LazySeg(self, pan._k)
        return seg

    @_locked
    def lazy_apply(self, seg, matchers):
        """apply_matchers on a handle: True when recorded, False when it has to run now"""
        from .matcher import RLEMatcher
        matchers = list(matchers)
        ok = self.fwd_lazy and self.uniform and self.bwd is None and seg._real is None and not seg._matched \
            and seg._k == self.n_matched
        if ok and self.n_matched == 0:
            things = set(self.rle_args[2]) & set(self.rle_args[0])
            ok = (all(type(mt) is RLEMatcher and mt.target_rle is None and mt.assign_new for mt in matchers)
                  and sorted(mt.class_id for mt in matchers) == sorted(things)
                  and all(mt.label_divisor == self.rle_args[1] for mt in matchers)
                  and len({(mt.merge_iou_thr, mt.merge_ioa_thr) for mt in matchers}) <= 1)
            if ok:
                self.matchers = matchers
                self.matcher_init = [(mt.next_label, mt.assign_new) for mt in matchers]
                for mt in matchers:
                    mt.target_rle = _Pending(self, 'fwd', mt)
        elif ok:
            ok = len(matchers) == len(self.matchers) and all(a is b for a, b in zip(matchers, self.matchers)) and all(
                isinstance(mt.target_rle, _Pending) and mt.target_rle.session is self for mt in matchers)
        if ok:
            seg._matched = True
            self.n_matched += 1
            return True
        # compute now: first everything recorded so far, then this slice through the matchers as they are
        if self.matchers is not None and any(a is b for a in matchers for b in self.matchers):
            self._forward_now()
        return False

    @_locked
    def note_applied_now(self, seg):
        """apply_matchers ran on the spot for a slice of this stack (after _forward_now brought the matchers up to date)"""
        if self.bwd is None and not self.fwd_lazy and not seg._matched:
            seg._matched = True
            self.n_matched += 1
            self.eager_matched += 1

    @_locked
    def lazy_backward(self, stack, matchers, axis_len):
        matchers = list(matchers)
        n = self.n_emitted
        ok = (self.closed and self._stack_ok() and self.bwd is None and self.matchers is not None
              and n > 0 and self.n_matched == n and axis_len == n and len(stack) == n
              and all(stack[i] is self.segs.get(i) for i in range(n))
              and len(matchers) == len(self.matchers) and all(a is b for a, b in zip(matchers, self.matchers)))
        if ok:
            self.bwd = 'lazy'
            for mt in self.matchers:
                mt.target_rle, mt.assign_new = _Pending(self, 'bwd', mt), False
        return ok

    # ------------------------------------------------------------------ whole-stack evaluation
    @torch.no_grad()
    @_locked
    def pan_stack(self):
        if self._pan is None:
            assert self.closed and self.uniform
            self._flush(self.calls)
            heads = self._consolidate()
            pan, emitted = self.engine._deferred_stack(heads, self.upsampling)
            assert len(emitted) == self.n_emitted
            if self.size is not None:
                pan = pan[:, :self.size[0], :self.size[1]].contiguous()
            self._pan = pan
        return self._pan

    def _consolidate(self):
        """the batches' head tensors as ONE tensor per head, copied batch by batch and freed as they go (a torch.cat
        would hold the stack twice: 137 GB of heads for 1024 slices of 2048^2 with five classes); the session keeps
        the result as its only chunk, so that values on demand still find their slice"""
        if len(self.chunks) == 1 and self.chunks[0][0] == 0 and self.chunks[0][1] == self.n_fwd:
            return self.chunks[0][2]
        heads = {}
        for k in _HEADS:
            first = self.chunks[0][2][k]
            out = torch.empty((self.n_fwd,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
            for lo, hi, part in self.chunks:
                out[lo:hi].copy_(part[k])
                part[k] = None
            heads[k] = out
        self.chunks = [(0, self.n_fwd, heads)]
        self.chunk_of = [0] * self.n_fwd
        return heads

    @_locked
    def tracker_instances(self, axis, shape3d, class_id):
        """the finished tracker of one class: what update_trackers over the backward pass + finish leave"""
        from . import device_tracks as DT
        from . import patterns
        key = (axis, tuple(int(s) for s in shape3d))
        if key not in self._tracks:
            labels, div, things, _ = self.rle_args
            if self._tables is None:
                pan = self.pan_stack()
                self._tables = patterns.tables_from_stack(pan, labels, things, div)
                mt = self.matchers[0] if self.matchers else None
                thr = (mt.merge_iou_thr, mt.merge_ioa_thr) if mt is not None else (0.25, 0.25)
                self._chain = patterns.chain_from_tables(self._tables[1], pan.shape[0], labels, things, div, *thr)
            table, host = self._tables
            tracks = DT.plane_tracks(table, host, self._chain[0], self._chain[1], axis, key[1], labels, div)
            self._tracks[key] = {tr.class_id: tr.instances for tr in tracks.trackers()}
        return self._tracks[key][class_id]


# ----------------------------------------------------------------------------------------------- handles
def _forcing(name):
    def method(self, *args, **kwargs):
        return getattr(self._force(), name)(*args, **kwargs)
    method.__name__ = name
    return method


class LazyPan:
    """The (1, 1, H, W) int64 panoptic image of one slice, not computed yet.  ``squeeze() / cpu() / numpy() / detach()
    / contiguous()`` return handles; anything else computes the tensor (per-slice engine code) and acts on it."""

    def __init__(self, session, k, ops=()):
        self._s, self._k, self._ops, self._val = session, k, ops, None

    def _chain(self, name, args):
        return LazyPan(self._s, self._k, self._ops + ((name, args),))

    def squeeze(self, *args):
        return self._chain('squeeze', args)

    def cpu(self):
        return self._chain('cpu', ())

    def numpy(self):
        return self._chain('numpy', ())

    def detach(self):
        return self._chain('detach', ())

    def contiguous(self):
        return self._chain('contiguous', ())

    def _is2d(self):
        return ('squeeze', ()) in self._ops

    def _force(self):
        if self._val is None:
            v = self._s.force_pan(self._k)
            for name, args in self._ops:
                v = getattr(v, name)(*args)
            self._val = v
        return self._val

    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        return getattr(self._force(), name)

    def __array__(self, dtype=None, copy=None):
        v = self._force()
        a = v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        return a if dtype is None else a.astype(dtype, copy=False)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        unwrap = lambda x: x._force() if isinstance(x, LazyPan) else x
        return func(*tree_map(unwrap, tuple(args)), **tree_map(unwrap, dict(kwargs or {})))

    def __reduce__(self):
        v = self._force()
        if isinstance(v, torch.Tensor):
            return (torch.from_numpy, (v.cpu().numpy(),))
        return (np.asarray, (v,))

    def __repr__(self):
        return f"LazyPan(slice {self._k}, {'computed' if self._val is not None else 'deferred'})"


for _n in ('__getitem__', '__len__', '__iter__', '__eq__', '__ne__', '__lt__', '__le__', '__gt__', '__ge__', '__add__',
           '__sub__', '__mul__', '__floordiv__', '__truediv__', '__mod__', '__and__', '__or__', '__xor__', '__radd__',
           '__rsub__', '__rmul__', '__neg__', '__invert__', '__bool__', '__int__', '__float__', '__index__'):
    setattr(LazyPan, _n, _forcing(_n))
LazyPan.__hash__ = object.__hash__


class _LazyMapping:
    """dict interface over a value that is computed on first use"""

    def _force(self):
        raise NotImplementedError

    def __reduce__(self):
        return (dict, (dict(self._force()),))

    def __repr__(self):
        return f"{type(self).__name__}(slice {self._k})"


for _n in ('__getitem__', '__setitem__', '__delitem__', '__iter__', '__len__', '__contains__', '__eq__', '__ne__',
           'keys', 'values', 'items', 'get', 'pop', 'popitem', 'setdefault', 'update', 'clear', 'copy'):
    setattr(_LazyMapping, _n, _forcing(_n))
_LazyMapping.__hash__ = None


class LazySeg(_LazyMapping):
    """rle_seg {class: {label: attrs}} of one slice (inference/rle.py), before or after forward matching"""

    def __init__(self, session, k):
        self._s, self._k, self._real, self._matched = session, k, None, False

    def _force(self):
        return self._s.force_seg(self._k)


class LazyFinal(_LazyMapping):
    """rle_seg of one slice after the backward pass (what backward_matching yields)"""

    def __init__(self, session, k):
        self._s, self._k = session, k

    def _force(self):
        return self._s.force_final(self._k)

    def __getitem__(self, class_id):
        if self._s.bwd == 'lazy':
            return LazyClass(self._s, self._k, class_id)
        return self._force()[class_id]


class LazyClass(_LazyMapping):
    """{label: attrs} of one class of one backward-matched slice (what InstanceTracker.update receives)"""

    def __init__(self, session, k, class_id):
        self._s, self._k, self._c = session, k, class_id

    def _force(self):
        return self._s.force_final(self._k)[self._c]


class LazyInstances(dict):
    """``InstanceTracker.instances`` while the tracker only holds handles.  finish() replaces it by the finished
    instances of the whole-stack path; any look inside before that files the recorded updates one by one."""

    def __init__(self, tracker):
        super().__init__()
        self._tracker, self._updates, self._resolved = tracker, [], False

    def record(self, handle, index2d):
        self._updates.append((handle, index2d))

    def fast_ok(self):
        if self._resolved or not self._updates:
            return None
        s = self._updates[0][0]._s
        n = s.n_emitted
        ok = (s.bwd == 'lazy' and len(self._updates) == n
              and all(h._s is s and h._c == self._tracker.class_id and h._k == i == n - 1 - j
                      for j, (h, i) in enumerate(self._updates)))
        return s if ok else None

    def resolve(self):
        if not self._resolved:
            self._resolved = True
            updates, self._updates = self._updates, []
            for handle, index2d in updates:
                self._tracker._update_now(handle._force(), index2d)

    def fill(self, final):
        self._resolved, self._updates = True, []
        dict.update(self, final)


def _resolving(name):
    def method(self, *args, **kwargs):
        self.resolve()
        return getattr(dict, name)(self, *args, **kwargs)
    method.__name__ = name
    return method


for _n in ('__getitem__', '__setitem__', '__delitem__', '__iter__', '__len__', '__contains__', '__eq__', '__ne__',
           '__repr__', 'keys', 'values', 'items', 'get', 'pop', 'popitem', 'setdefault', 'update', 'clear', 'copy'):
    setattr(LazyInstances, _n, _resolving(_n))
LazyInstances.__reduce__ = lambda self: (dict, (dict(self.items()),))
