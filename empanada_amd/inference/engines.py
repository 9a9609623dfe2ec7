"""Inference engines with the reference's class names, constructor arguments and call protocol
(``empanada/inference/engines.py``): the model forward stays in PyTorch-ROCm, everything after the
probabilities runs in libemp_hip.so.

Per-slice protocol (drop-in): ``engine(image[, size, upsampling]) -> pan | None`` and
``engine.end() -> list`` exactly like engines.py:141-159, 183-221, 300-325, 351-394.
Whole-stack protocol (MI355X-native fast path): ``engine.infer_stack(...)`` batches slices through the
model and ``postprocess.panoptic_stack`` turns the resident head tensors into labels in five
kernel groups; results are identical to feeding the slices one by one.
"""
import math
import os
from collections import deque

import torch
import torch.nn.functional as F

from .. import _hip
from .deferred import StackSession
from .postprocess import (factor_pad, find_instance_center, get_panoptic_segmentation, group_pixels,
                          merge_semantic_and_instance, panoptic_stack)

__all__ = ['PanopticDeepLabEngine', 'PanopticDeepLabEngine3d', 'PanopticDeepLabRenderEngine',
           'PanopticDeepLabRenderEngine3d', 'MultiGPUInferenceEngine', 'logits_to_prob']


@torch.no_grad()
def logits_to_prob(logits):
    """engines.py:22-30.  fp32 logits on the GPU go through emp_logits_to_prob (D2); anything else (a host tensor, a
    half-precision model) keeps the library call the reference makes."""
    if logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4:
        return _hip.logits_to_prob(logits)
    if logits.size(1) > 1:
        return F.softmax(logits, dim=1)
    return torch.sigmoid(logits)


def _on_gpu_f32(t):
    """fp32 view of a head tensor on the GPU (the kernels take device pointers; there is no host path)"""
    _hip.require_gpu()
    return t.float() if t.is_cuda else t.float().cuda()


class _Engine:
    """Holds the model in eval mode (engines.py:32-45); subclasses define infer / __call__."""

    def __init__(self, model):
        self.model = model.eval()

    def to_model_device(self, tensor):
        return tensor.to(next(self.model.parameters()).device, non_blocking=True)

    def infer(self, image):
        raise NotImplementedError

    def __call__(self, image):
        raise NotImplementedError


class _MedianQueue:
    """Recursive median over the last ``median_kernel_size`` engine outputs (engines.py:47-90).

    ``median_queue`` holds the output dicts of the most recent slices.  While it holds at most ks // 2 + 1 items the
    newest one is handed out unfiltered; then nothing until the queue is full; from then on the item in the middle is
    handed out AFTER its tensors under ``keys`` were replaced, in the queue itself, by the per-pixel median over the
    whole queue (``emp_median_step``) -- later medians therefore see already-filtered values on their left side.
    ``end()`` returns the ks // 2 items to the right of the middle, unfiltered."""

    def __init__(self, median_kernel_size, **kwargs):
        super().__init__(**kwargs)
        assert median_kernel_size % 2 == 1, "Kernel size must be odd integer!"
        assert median_kernel_size <= _hip.MAX_KS, f"median kernel sizes above {_hip.MAX_KS} are not supported"
        self.ks = median_kernel_size
        self.mid_idx = median_kernel_size // 2
        self.reset()

    def reset(self):
        self.median_queue = deque(maxlen=self.ks)
        self._session = None

    # ---- deferred evaluation (inference/deferred.py) ----------------------------------------------------------
    def _init_deferred(self, deferred, batch):
        """deferred=None reads EMP_DEFERRED (default off: every call computes on the spot, like the reference)"""
        if deferred is None:
            deferred = os.environ.get('EMP_DEFERRED', '0') == '1'
        self.deferred, self.deferred_batch = bool(deferred), int(batch)

    def _open_session(self):
        """the session the next image belongs to; an engine that is called again after end() without reset()
        continues, like the reference's, with the queue as end() left it -- computed on the spot from then on"""
        s = self._session
        if s is not None and s.closed:
            s.go_eager()
            self.deferred, self._session, s = False, None, None
        elif s is None:
            s = self._session = StackSession(self, self.deferred_batch)
        return s

    def _end_deferred(self):
        s = self._session
        return [] if s is None else s.end()

    def enqueue(self, item):
        self.median_queue.append(item)

    @torch.no_grad()
    def get_median(self, key):
        return _hip.median_step([_on_gpu_f32(item[key]) for item in self.median_queue])

    def get_next(self, keys):
        held = len(self.median_queue)
        if held == self.ks:
            middle = self.median_queue[self.mid_idx]
            middle.update({key: self.get_median(key) for key in keys})
            return middle
        return self.median_queue[-1] if held <= self.mid_idx else None

    def end(self):
        return [item for pos, item in enumerate(self.median_queue) if pos > self.mid_idx]


def _check_single_image(image):
    assert image.ndim == 4 and image.size(0) == 1          # engines.py:144,203,306,369: one image per call


class PanopticDeepLabEngine(_Engine):
    """Full-resolution heads, one image per call -> panoptic labels (engines.py:92-159)."""

    def __init__(self, model, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1,
                 nms_kernel=7, confidence_thr=0.5, **kwargs):
        super().__init__(model=model)
        self.thing_list, self.label_divisor = thing_list, label_divisor
        self.stuff_area, self.void_label = stuff_area, void_label
        self.nms_threshold, self.nms_kernel = nms_threshold, nms_kernel
        self.confidence_thr = confidence_thr

    @torch.no_grad()
    def _harden_seg(self, sem):
        """probabilities (N,C,H,W) -> class map (N,1,H,W) int64: argmax over C > 1 channels, ``p >= confidence_thr``
        for C = 1 (engines.py:114-121; emp_harden)."""
        sem = _on_gpu_f32(sem)
        N, C, H, W = sem.shape
        hard = torch.empty((N, H, W), dtype=torch.uint8, device=sem.device)
        for n in range(N):
            plane = sem[n:n + 1].contiguous()
            _hip.call('emp_harden', plane.data_ptr(), 1, C, H * W, float(self.confidence_thr), hard[n].data_ptr(),
                      _hip.stream())
        return hard[:, None].long()

    @torch.no_grad()
    def infer(self, image):
        """model outputs plus 'sem' = class probabilities (the logits stay under 'sem_logits'), engines.py:123-129"""
        heads = self.model(image)
        heads['sem'] = logits_to_prob(heads['sem_logits'])
        return heads

    @torch.no_grad()
    def postprocess(self, sem, ctr_hmp, offsets):
        return get_panoptic_segmentation(sem, ctr_hmp, offsets, self.thing_list, self.label_divisor, self.stuff_area,
                                         self.void_label, self.nms_threshold, self.nms_kernel)[0]

    def _stack_form_ok(self, heads):
        """one slice through the whole-stack kernels (postprocess.panoptic_stack with a median of 1: five launch groups,
        one host sync) instead of the reference-shaped functions below (~40 small launches and six syncs per slice).
        Same labels (tests/test_pipeline_gpu.py pins the two forms to each other and to the reference's fixtures); only
        when nobody overrode the functions it stands in for."""
        cls = type(self)
        mine = (PanopticDeepLabEngine, PanopticDeepLabEngine3d, PanopticDeepLabRenderEngine, PanopticDeepLabRenderEngine3d)
        owner = lambda name: next(c for c in cls.__mro__ if name in c.__dict__)
        return (len(self.thing_list) > 0
                and all(owner(n) in mine for n in ('postprocess', '_harden_seg', 'get_instance_cells', 'get_panoptic_seg')
                        if hasattr(cls, n))
                and all(isinstance(heads[k], torch.Tensor) and heads[k].is_cuda and heads[k].dim() == 4
                        and heads[k].size(0) == 1 for k in ('sem', 'ctr_hmp', 'offsets')))

    def _labels_of(self, heads):
        if self._stack_form_ok(heads):
            params = dict(self._stack_params(), median_kernel_size=1)
            pan, _ = panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'], coarse_boundaries=False,
                                    upsampling=1, out_dtype=torch.int64, **params)
            return pan[None]
        return self.postprocess(self._harden_seg(heads['sem']), heads['ctr_hmp'], heads['offsets'])

    def __call__(self, image):
        _check_single_image(image)
        return self._labels_of(self.infer(self.to_model_device(image)))

    # ---- MI355X-native whole-stack path ----------------------------------------------------
    def _stack_params(self):
        return dict(thing_list=self.thing_list, label_divisor=self.label_divisor, stuff_area=self.stuff_area,
                    void_label=self.void_label, nms_threshold=self.nms_threshold, nms_kernel=self.nms_kernel,
                    confidence_thr=self.confidence_thr, median_kernel_size=getattr(self, 'ks', 1))

    @torch.no_grad()
    def forward_stack(self, images, batch_size=16, model_args=()):
        """Run the model over (D,1,H,W) images in batches; returns resident head tensors
        {'sem' (D,C,Hp,Wp) probabilities, 'ctr_hmp', 'offsets'}."""
        outs = {'sem': [], 'ctr_hmp': [], 'offsets': []}
        for s in range(0, images.size(0), batch_size):
            x = self.to_model_device(images[s:s + batch_size])
            o = self.model(x, *model_args)
            outs['sem'].append(logits_to_prob(o['sem_logits']).float())
            outs['ctr_hmp'].append(o['ctr_hmp'].float())
            outs['offsets'].append(o['offsets'].float())
        return {k: torch.cat(v, dim=0) for k, v in outs.items()}

    @torch.no_grad()
    def postprocess_stack(self, heads, coarse_boundaries=False, upsampling=1):
        return panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'], coarse_boundaries=coarse_boundaries,
                              upsampling=upsampling, **self._stack_params())


class PanopticDeepLabEngine3d(_MedianQueue, PanopticDeepLabEngine):
    """PanopticDeepLabEngine behind the recursive median queue: ``engine(image)`` returns None while the queue
    fills, ``end()`` flushes the last ks // 2 slices (engines.py:161-221)."""

    def __init__(self, model, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1,
                 nms_kernel=7, confidence_thr=0.5, median_kernel_size=3, deferred=None, deferred_batch=16, **kwargs):
        super().__init__(model=model, thing_list=thing_list, label_divisor=label_divisor, stuff_area=stuff_area,
                         void_label=void_label, nms_threshold=nms_threshold, nms_kernel=nms_kernel,
                         confidence_thr=confidence_thr, median_kernel_size=median_kernel_size, **kwargs)
        self._init_deferred(deferred, deferred_batch)

    def __call__(self, image):
        _check_single_image(image)
        if self.deferred:
            session = self._open_session()
            if session is not None:
                return session.add(self.to_model_device(image))
        self.enqueue(self.infer(self.to_model_device(image)))
        ready = self.get_next(keys=['sem'])
        return None if ready is None else self._labels_of(ready)

    def end(self):
        return self._end_deferred() if self.deferred else self._end_now()

    def _end_now(self, upsampling=1):
        tail = _MedianQueue.end(self)
        for heads in tail:                                 # the reference leaves the hardened map in the queue item
            heads['sem'] = self._harden_seg(heads['sem'])
        return [self.postprocess(heads['sem'], heads['ctr_hmp'], heads['offsets']) for heads in tail]

    # what a deferred session calls
    def _labels_now(self, ready, upsampling=1):
        return self._labels_of(ready)

    def _deferred_infer(self, x, upsampling=1):
        return self.infer(x)

    def _deferred_stack(self, heads, upsampling=1):
        return self.postprocess_stack(heads, coarse_boundaries=False, upsampling=1)


class PanopticDeepLabRenderEngine(PanopticDeepLabEngine):
    """PointRend models: instance heads at 1/4 resolution (``coarse_boundaries``), semantic head rendered at
    ``upsampling`` x the input; images are padded to ``padding_factor`` and the result cropped back to ``size``
    (engines.py:223-325)."""

    def __init__(self, model, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1,
                 nms_kernel=7, confidence_thr=0.5, padding_factor=16, coarse_boundaries=True, **kwargs):
        super().__init__(model=model, thing_list=thing_list, label_divisor=label_divisor, stuff_area=stuff_area,
                         void_label=void_label, nms_threshold=nms_threshold, nms_kernel=nms_kernel,
                         confidence_thr=confidence_thr)
        self.padding_factor = padding_factor
        self.coarse_boundaries = coarse_boundaries

    @torch.no_grad()
    def infer(self, image, render_steps=2):
        heads = self.model(image, render_steps, interpolate_ins=not self.coarse_boundaries)
        heads['sem'] = logits_to_prob(heads['sem_logits'])
        return heads

    @torch.no_grad()
    def get_instance_cells(self, ctr_hmp, offsets, upsampling=1):
        """centre NMS + nearest-centre vote on the (possibly 1/4-resolution) instance heads, enlarged by nearest
        neighbour to the resolution of the semantic map -> (1,1,H,W) fp32 ids, 0 everywhere without centres
        (engines.py:257-275)."""
        step = 4 if self.coarse_boundaries else 1
        centres = find_instance_center(ctr_hmp, self.nms_threshold, self.nms_kernel)
        if centres.size(0) > 0:
            cells = group_pixels(centres, offsets, step=step).float()[None]
        else:
            cells = torch.zeros_like(ctr_hmp if ctr_hmp.is_cuda else ctr_hmp.cuda())
        return F.interpolate(cells, scale_factor=int(upsampling * step), mode='nearest')

    @torch.no_grad()
    def get_panoptic_seg(self, sem, instance_cells):
        """cells masked to the thing classes, then the majority-class fusion (engines.py:277-292)"""
        is_thing = torch.zeros_like(sem)
        for cls in self.thing_list:
            is_thing[sem == cls] = 1
        ids = (is_thing * instance_cells[0]).long()
        return merge_semantic_and_instance(sem, ids, self.label_divisor, self.thing_list, self.stuff_area,
                                           self.void_label)

    @torch.no_grad()
    def postprocess(self, sem, instance_cells):
        return self.get_panoptic_seg(self._harden_seg(sem)[0], instance_cells)

    @staticmethod
    def _render_steps(upsampling):
        extra = math.log(upsampling, 2)
        assert extra.is_integer(), "Upsampling factor not log base 2!"
        return int(2 + extra)

    def _padded_heads(self, image, upsampling):
        """pad, move to the model's device, forward with 2 + log2(upsampling) render steps"""
        steps = self._render_steps(upsampling)
        _check_single_image(image)
        return self.infer(self.to_model_device(factor_pad(image, self.padding_factor)), steps)

    def _cropped_labels(self, heads, size, upsampling):
        if self._stack_form_ok(heads):
            params = dict(self._stack_params(), median_kernel_size=1)
            pan, _ = panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'],
                                    coarse_boundaries=self.coarse_boundaries, upsampling=upsampling,
                                    out_dtype=torch.int64, **params)
            return pan[..., :size[0], :size[1]]          # (1, h, w) like get_panoptic_seg on sem[0] (engines.py:277-292)
        cells = self.get_instance_cells(heads['ctr_hmp'], heads['offsets'], upsampling)
        return self.postprocess(heads['sem'], cells)[..., :size[0], :size[1]]

    def __call__(self, image, size, upsampling=1):
        return self._cropped_labels(self._padded_heads(image, upsampling), size, upsampling)

    @torch.no_grad()
    def postprocess_stack(self, heads, upsampling=1):
        return panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'],
                              coarse_boundaries=self.coarse_boundaries, upsampling=upsampling, **self._stack_params())


class PanopticDeepLabRenderEngine3d(_MedianQueue, PanopticDeepLabRenderEngine):
    """PanopticDeepLabRenderEngine behind the recursive median queue; every queue item remembers the ``size`` it
    must be cropped to (engines.py:327-394)."""

    def __init__(self, model, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1,
                 nms_kernel=7, confidence_thr=0.5, median_kernel_size=3, padding_factor=16, coarse_boundaries=True,
                 deferred=None, deferred_batch=16, **kwargs):
        super().__init__(model=model, thing_list=thing_list, label_divisor=label_divisor, stuff_area=stuff_area,
                         void_label=void_label, nms_threshold=nms_threshold, nms_kernel=nms_kernel,
                         confidence_thr=confidence_thr, median_kernel_size=median_kernel_size,
                         padding_factor=padding_factor, coarse_boundaries=coarse_boundaries)
        self._init_deferred(deferred, deferred_batch)

    def __call__(self, image, size, upsampling=1):
        if self.deferred:
            session = self._open_session()
            if session is not None:
                self._render_steps(upsampling)
                _check_single_image(image)
                return session.add(self.to_model_device(factor_pad(image, self.padding_factor)), size, upsampling)
        heads = self._padded_heads(image, upsampling)
        heads['size'] = size
        self.enqueue(heads)
        ready = self.get_next(keys=['sem'])
        return None if ready is None else self._cropped_labels(ready, ready['size'], upsampling)

    def end(self, upsampling=1):
        return self._end_deferred() if self.deferred else self._end_now(upsampling)

    def _end_now(self, upsampling=1):
        return [self._cropped_labels(heads, heads['size'], upsampling) for heads in _MedianQueue.end(self)]

    # what a deferred session calls
    def _labels_now(self, ready, upsampling=1):
        return self._cropped_labels(ready, ready['size'], upsampling)

    def _deferred_infer(self, x, upsampling=1):
        return self.infer(x, self._render_steps(upsampling))

    def _deferred_stack(self, heads, upsampling=1):
        return self.postprocess_stack(heads, upsampling)


class MultiGPUInferenceEngine(PanopticDeepLabRenderEngine):
    """The engine `scripts/inference3d_multigpu.py:289,350-361` instantiates but the reference never
    defines: `.infer(image) -> {'sem' (probabilities), 'ctr_hmp', 'offsets'}` and
    `.get_instance_cells(ctr_hmp, offsets)` on full-resolution heads."""

    def __init__(self, model, thing_list=(1,), **engine_params):
        engine_params.setdefault('coarse_boundaries', False)
        super().__init__(model=model, thing_list=list(thing_list), **engine_params)

    @torch.no_grad()
    def infer(self, image, render_steps=2):
        model_out = self.model(self.to_model_device(image))
        model_out['sem'] = logits_to_prob(model_out['sem_logits'])
        return model_out
