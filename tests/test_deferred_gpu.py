"""GPU: the per-slice protocol with a DEFERRED 3d engine (empanada_amd/inference/deferred.py) gives the values of the
same protocol computed on the spot -- when nobody looks inside a handle (whole-stack path under the per-slice names) and
when somebody does, at any stage (replay through the per-slice functions).  Reference call sequence:
scripts/pdl_inference3d.py:163-198."""
import pickle

import numpy as np
import pytest
import torch

from conftest import assert_instances_equal, load_golden
from empanada_amd import synthetic as SY
from test_pipeline_gpu import _engine_case

pytestmark = pytest.mark.gpu


class BatchStub(torch.nn.Module):
    """hands out pre-computed head tensors for as many slices as the batch holds; 'sem_logits' = probabilities"""

    def __init__(self, heads, logits=False):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.heads, self.t, self.batches, self.logits = heads, 0, [], logits

    def forward(self, x, *a, **k):
        n = x.size(0)
        o = {k2: v[self.t:self.t + n].clone().to(self.p.device) for k2, v in self.heads.items()}
        p = o.pop('sem')
        if self.logits:                                    # sigmoid / softmax of these give the planted probabilities back
            p = p.clamp(1e-6, 1 - 1e-6)
            p = torch.log(p / (1 - p)) if p.size(1) == 1 else torch.log(p)
        o['sem_logits'] = p
        self.t += n
        self.batches.append(n)
        return o


@pytest.fixture()
def prob_passthrough(monkeypatch):
    from empanada_amd.inference import engines
    monkeypatch.setattr(engines, 'logits_to_prob', lambda x: x)
    return engines


@pytest.mark.parametrize('look', ['after_end', 'at_once'])
def test_deferred_engines_hand_out_the_reference_images(prob_passthrough, look):
    """engines.npz (outputs of the reference's four 3d engine configurations) through deferred engines: the handles
    hold the reference's images whether they are read after end() or the moment they are handed out"""
    EN = prob_passthrough
    from empanada_amd.inference.deferred import LazyPan
    g = load_golden('engines')
    for i in range(int(g['n'])):
        heads, kw, coarse, render, exp = _engine_case(g, i)
        S, _, H, W = heads['sem'].shape
        stub = BatchStub(heads).cuda()
        if render:
            eng = EN.PanopticDeepLabRenderEngine3d(stub, padding_factor=16, coarse_boundaries=coarse, deferred=True,
                                                   deferred_batch=4, **kw)
            call = lambda: eng(torch.zeros(1, 1, H, W), (H - 3, W - 5))
        else:
            eng = EN.PanopticDeepLabEngine3d(stub, deferred=True, deferred_batch=4, **kw)
            call = lambda: eng(torch.zeros(1, 1, H, W))
        outs = []
        for t in range(S):
            o = call()
            if o is not None:
                assert isinstance(o, LazyPan)
                outs.append(np.asarray(o.cpu().numpy()) if look == 'at_once' else o.cpu().numpy())
        outs += [o.cpu().numpy() for o in eng.end()]
        got = np.stack([np.asarray(o) for o in outs])
        assert got.dtype == np.int64
        np.testing.assert_array_equal(got, exp, err_msg=f'engine case {i}')
        if look == 'after_end':
            assert max(stub.batches) == min(4, S)          # the forward really ran in batches


def _protocol(heads, axis, shape3d, labels, thing, ks, deferred, look=None, batch=5, crop=None):
    """scripts/pdl_inference3d.py:163-198 for one plane; look = where somebody reads a handle; crop = (h, w): the
    PointRend engine (1/4-resolution instance heads, padded input, cropped output) instead of the plain one"""
    from empanada_amd.inference import engines as EN
    from empanada_amd.inference import patterns as PA
    from empanada_amd.inference import rle
    S, _, H, W = heads['sem'].shape
    kw = dict(thing_list=thing, label_divisor=1000, stuff_area=16, void_label=0, nms_threshold=0.1, nms_kernel=7,
              confidence_thr=0.5, median_kernel_size=ks, deferred=deferred, deferred_batch=batch)
    if crop is None:
        engine = EN.PanopticDeepLabEngine3d(BatchStub(heads, logits=True).cuda(), **kw)
        eng = lambda image: engine(image)
    else:
        engine = EN.PanopticDeepLabRenderEngine3d(BatchStub(heads, logits=True).cuda(), padding_factor=16,
                                                  coarse_boundaries=True, **kw)
        eng = lambda image: engine(image[..., :crop[0], :crop[1]], crop)
    eng.end = engine.end
    matchers = PA.create_matchers(thing, 1000, 0.25, 0.25)
    trackers = PA.create_axis_trackers({axis: 0}, labels, 1000, shape3d)[axis]
    stack, seen = [], {}

    import multiprocessing
    q = multiprocessing.Queue() if look == 'queue' else None

    def consume(pan):
        n = len(stack)
        if look == 'queue':                                # an mp.Queue pickles in its feeder THREAD, beside the engine calls
            q.put(pan.squeeze().cpu().numpy())
            pan = q.get(timeout=60)
            assert isinstance(pan, np.ndarray)
        elif look == 'script':                               # the reference script: a numpy array goes into an mp.Queue
            pan = pickle.loads(pickle.dumps(pan.squeeze().cpu().numpy()))
            assert isinstance(pan, np.ndarray)
        elif look != 'queue':
            pan = pan.squeeze().cpu().numpy()
        if look == 'pan' and n == 3:
            seen['pan'] = np.asarray(pan).copy()
        seg = rle.pan_seg_to_rle_seg(pan, labels, 1000, thing, True)
        if look == 'rle' and n == 2:
            seen['rle'] = sorted(seg[thing[0]].keys())
        seg = PA.apply_matchers(seg, matchers)
        if look == 'seg' and n == 4:
            seen['seg'] = sorted(seg[thing[0]].keys())
            seen['next_label'] = matchers[0].next_label
        if look == 'pickle_matcher' and n == 4:            # matchers cross mp.Queues in the reference's scripts
            clone = pickle.loads(pickle.dumps(matchers[0]))
            seen['clone'] = [clone.next_label] + sorted(clone.target_rle.keys())
        stack.append(seg)

    for t in range(S):
        pan = eng(torch.zeros(1, 1, H, W))
        if pan is not None:
            consume(pan)
    if look != 'no_end':                                   # a caller that forgets end(): the stack is not closed
        for pan in eng.end():
            consume(pan)
    for idx, rs in PA.backward_matching(stack, matchers, len(stack)):
        if look == 'final' and idx == len(stack) - 3:
            seen['final'] = sorted(rs[thing[0]].keys())
        PA.update_trackers(rs, idx, trackers)
        if look == 'instances' and idx == 2:
            seen['instances'] = sorted(trackers[0].instances.keys())
    PA.finish_tracking(trackers)
    return trackers, seen, engine


@pytest.mark.parametrize('axis', ['xy', 'xz', 'yz'])
@pytest.mark.parametrize('C', [1, 3])
def test_deferred_protocol_fills_the_trackers_like_the_per_slice_protocol(axis, C):
    from empanada_amd.inference.deferred import LazyInstances
    shape = (22, 40, 44)
    lab, cls = SY.planted_labels(shape, fill=0.2, rmin=4, rmax=9, seed=5, n_classes=C)
    heads = SY.planted_heads(lab, cls, axis, n_classes=C, seed=7, coarse=False)
    thing = [1] if C == 1 else [1, 2]                      # C = 3: two thing classes and a stuff class
    labels = [1] if C == 1 else [1, 2, 3]
    exp_full, _, _ = _protocol(heads, axis, shape, labels, thing, 5, deferred=False)
    assert sum(len(t.instances) for t in exp_full) > 5
    for look in (None, 'pan', 'rle', 'seg', 'final', 'instances', 'script', 'no_end', 'pickle_matcher', 'queue'):
        got, seen, eng = _protocol(heads, axis, shape, labels, thing, 5, deferred=True, look=look)
        ref = _protocol(heads, axis, shape, labels, thing, 5, deferred=False, look=look) if look else (exp_full, {})
        exp, ref_seen = ref[0], ref[1]
        for k in ref_seen:
            np.testing.assert_array_equal(np.asarray(seen[k]), np.asarray(ref_seen[k]), err_msg=f'{look}:{k}')
        for a, b in zip(got, exp):
            assert a.finished and not isinstance(a.instances, LazyInstances)
            assert_instances_equal(a.instances, b.instances)
        s = eng._session
        if look is None:                                   # nobody looked: no per-slice image was ever formed
            assert s._pan is not None and not s.eager_out and s.bwd == 'lazy'
        if look == 'script':                               # everybody looked at once: nothing was left to the stack path
            assert s._pan is None and len(s.eager_out) == s.n_emitted


def test_deferred_protocol_with_the_pointrend_engine():
    """PanopticDeepLabRenderEngine3d: padded input, 1/4-resolution instance heads, output cropped to `size`"""
    shape = (14, 48, 64)
    crop = (45, 59)
    lab, cls = SY.planted_labels(shape, fill=0.25, rmin=4, rmax=10, seed=11)
    heads = SY.planted_heads(lab, cls, 'xy', seed=3, coarse=True)
    shape3d = (shape[0],) + crop
    exp, _, _ = _protocol(heads, 'xy', shape3d, [1], [1], 3, deferred=False, crop=crop)
    assert len(exp[0].instances) > 3
    for look in (None, 'pan', 'seg'):
        got, _, eng = _protocol(heads, 'xy', shape3d, [1], [1], 3, deferred=True, look=look, crop=crop)
        assert_instances_equal(got[0].instances, exp[0].instances)
        if look is None:
            assert eng._session._pan is not None and tuple(eng._session._pan.shape[1:]) == crop


def test_deferred_engine_falls_back_for_slices_of_different_sizes_and_reuse_after_end():
    """a stack whose slices change shape has no whole-stack form: values come from the per-slice code; an engine that
    is called again after end() continues with the queue as end() left it, like the reference's"""
    from empanada_amd.inference import engines as EN
    kw = dict(thing_list=[1], label_divisor=1000, stuff_area=16, void_label=0, nms_threshold=0.1, nms_kernel=7,
              confidence_thr=0.5, median_kernel_size=3)
    lab, cls = SY.planted_labels((6, 48, 48), fill=0.2, rmin=4, rmax=8, seed=3)
    heads = SY.planted_heads(lab, cls, 'xy', seed=1)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.t = 0

        def forward(self, x, *a, **k):
            n, (h, w) = x.size(0), x.shape[-2:]
            o = {k2: v[self.t:self.t + n, :, :h, :w].clone().cuda() for k2, v in heads.items()}
            p = o.pop('sem').clamp(1e-6, 1 - 1e-6)
            o['sem_logits'] = torch.log(p / (1 - p))
            self.t += n
            return o

    def run(deferred):
        eng = EN.PanopticDeepLabRenderEngine3d(Net().cuda(), padding_factor=16, coarse_boundaries=False,
                                               deferred=deferred, **kw)
        outs = []
        sizes = [(48, 48), (48, 48), (40, 45), (48, 48)]      # all pad to 48 x 48; the crop differs
        for t, sz in enumerate(sizes):
            o = eng(torch.zeros(1, 1, *sz), sz)
            outs.append(None if o is None else np.asarray(o.cpu().numpy()))
        outs += [np.asarray(o.cpu().numpy()) for o in eng.end()]
        for t in range(2):                                 # reuse without reset()
            o = eng(torch.zeros(1, 1, 48, 48), (48, 48))
            outs.append(None if o is None else np.asarray(o.cpu().numpy()))
        return outs

    a, b = run(False), run(True)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert (x is None) == (y is None)
        if x is not None:
            np.testing.assert_array_equal(x, y)
