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
from collections import deque

import torch
import torch.nn.functional as F

from .. import _hip
from .postprocess import (factor_pad, find_instance_center, get_panoptic_segmentation, group_pixels,
                          merge_semantic_and_instance, panoptic_stack)

__all__ = ['PanopticDeepLabEngine', 'PanopticDeepLabEngine3d', 'PanopticDeepLabRenderEngine',
           'PanopticDeepLabRenderEngine3d', 'MultiGPUInferenceEngine', 'logits_to_prob']


@torch.no_grad()
def logits_to_prob(logits):
    """engines.py:22-30"""
    if logits.size(1) > 1:
        return F.softmax(logits, dim=1)
    return torch.sigmoid(logits)


class _Engine:
    """engines.py:32-45"""

    def __init__(self, model):
        self.model = model.eval()

    def infer(self, image):
        raise NotImplementedError

    def to_model_device(self, tensor):
        device = next(self.model.parameters()).device
        return tensor.to(device, non_blocking=True)

    def __call__(self, image):
        raise NotImplementedError


class _MedianQueue:
    """engines.py:47-90.  The deque holds the engine's output dicts; get_next(keys) overwrites the
    middle item's tensors with the per-pixel median over the queue (emp_median_step), which makes
    the filter recursive exactly like the reference."""

    def __init__(self, median_kernel_size, **kwargs):
        super().__init__(**kwargs)
        assert median_kernel_size % 2 == 1, "Kernel size must be odd integer!"
        assert median_kernel_size <= _hip.MAX_KS, f"median kernel sizes above {_hip.MAX_KS} are not supported"
        self.ks = median_kernel_size
        self.mid_idx = (median_kernel_size - 1) // 2
        self.median_queue = deque(maxlen=median_kernel_size)

    def reset(self):
        self.median_queue = deque(maxlen=self.ks)

    @torch.no_grad()
    def get_median(self, key):
        slices = [out[key] for out in self.median_queue]
        _hip.require_gpu()
        slices = [s.float().cuda() if not s.is_cuda else s.float() for s in slices]
        return _hip.median_step(slices)

    def get_next(self, keys):
        nq = len(self.median_queue)
        if nq <= self.mid_idx:
            output = self.median_queue[-1]
        elif nq < self.ks:
            return None
        else:
            output = self.median_queue[self.mid_idx]
            for key in keys:
                output[key] = self.get_median(key)
        return output

    def enqueue(self, item):
        self.median_queue.append(item)

    def end(self):
        return list(self.median_queue)[self.mid_idx + 1:]


class PanopticDeepLabEngine(_Engine):
    """engines.py:92-159"""

    def __init__(self, model, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1,
                 nms_kernel=7, confidence_thr=0.5, **kwargs):
        super().__init__(model=model)
        self.thing_list = thing_list
        self.label_divisor = label_divisor
        self.stuff_area = stuff_area
        self.void_label = void_label
        self.nms_threshold = nms_threshold
        self.nms_kernel = nms_kernel
        self.confidence_thr = confidence_thr

    @torch.no_grad()
    def _harden_seg(self, sem):
        """engines.py:114-121 -> (N,1,H,W) int64 (emp_harden)."""
        _hip.require_gpu()
        sem = sem.float().cuda() if not sem.is_cuda else sem.float()
        N, C, H, W = sem.shape
        out = torch.empty((N, H, W), dtype=torch.uint8, device=sem.device)
        for n in range(N):
            one = sem[n:n + 1].contiguous()
            _hip.call('emp_harden', one.data_ptr(), 1, C, H * W, float(self.confidence_thr),
                      out[n].data_ptr(), _hip.stream())
        return out[:, None].long()

    @torch.no_grad()
    def infer(self, image):
        model_out = self.model(image)
        model_out['sem'] = logits_to_prob(model_out['sem_logits'])   # notice that sem is NOT sem_logits
        return model_out

    @torch.no_grad()
    def postprocess(self, sem, ctr_hmp, offsets):
        pan_seg, _ = get_panoptic_segmentation(
            sem, ctr_hmp, offsets, self.thing_list, self.label_divisor, self.stuff_area, self.void_label,
            self.nms_threshold, self.nms_kernel)
        return pan_seg

    def __call__(self, image):
        assert image.ndim == 4 and image.size(0) == 1
        image = self.to_model_device(image)
        model_out = self.infer(image)
        model_out['sem'] = self._harden_seg(model_out['sem'])
        return self.postprocess(model_out['sem'], model_out['ctr_hmp'], model_out['offsets'])

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
    """engines.py:161-221"""

    def __init__(self, model, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1,
                 nms_kernel=7, confidence_thr=0.5, median_kernel_size=3, **kwargs):
        super().__init__(model=model, thing_list=thing_list, label_divisor=label_divisor, stuff_area=stuff_area,
                         void_label=void_label, nms_threshold=nms_threshold, nms_kernel=nms_kernel,
                         confidence_thr=confidence_thr, median_kernel_size=median_kernel_size, **kwargs)

    def end(self):
        final_segs = []
        for model_out in list(self.median_queue)[self.mid_idx + 1:]:
            model_out['sem'] = self._harden_seg(model_out['sem'])
            final_segs.append(self.postprocess(model_out['sem'], model_out['ctr_hmp'], model_out['offsets']))
        return final_segs

    def __call__(self, image):
        assert image.ndim == 4 and image.size(0) == 1
        image = self.to_model_device(image)
        model_out = self.infer(image)
        self.enqueue(model_out)
        median_out = self.get_next(keys=['sem'])
        if median_out is None:
            return None
        return self.postprocess(self._harden_seg(median_out['sem']), median_out['ctr_hmp'], median_out['offsets'])


class PanopticDeepLabRenderEngine(PanopticDeepLabEngine):
    """engines.py:223-325"""

    def __init__(self, model, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1,
                 nms_kernel=7, confidence_thr=0.5, padding_factor=16, coarse_boundaries=True, **kwargs):
        super().__init__(model=model, thing_list=thing_list, label_divisor=label_divisor, stuff_area=stuff_area,
                         void_label=void_label, nms_threshold=nms_threshold, nms_kernel=nms_kernel,
                         confidence_thr=confidence_thr)
        self.padding_factor = padding_factor
        self.coarse_boundaries = coarse_boundaries

    @torch.no_grad()
    def infer(self, image, render_steps=2):
        model_out = self.model(image, render_steps, interpolate_ins=not self.coarse_boundaries)
        model_out['sem'] = logits_to_prob(model_out['sem_logits'])
        return model_out

    @torch.no_grad()
    def get_instance_cells(self, ctr_hmp, offsets, upsampling=1):
        """engines.py:257-275 -> (1,1,H,W) fp32 cells."""
        ctr = find_instance_center(ctr_hmp, self.nms_threshold, self.nms_kernel)
        step = 4 if self.coarse_boundaries else 1
        if ctr.size(0) == 0:
            instance_cells = torch.zeros_like(ctr_hmp.cuda() if not ctr_hmp.is_cuda else ctr_hmp)
        else:
            instance_cells = group_pixels(ctr, offsets, step=step).float()[None]
        return F.interpolate(instance_cells, scale_factor=int(upsampling * step), mode='nearest')

    @torch.no_grad()
    def get_panoptic_seg(self, sem, instance_cells):
        """engines.py:277-292"""
        instance_seg = torch.zeros_like(sem)
        for thing_class in self.thing_list:
            instance_seg[sem == thing_class] = 1
        instance_seg = (instance_seg * instance_cells[0]).long()
        return merge_semantic_and_instance(sem, instance_seg, self.label_divisor, self.thing_list, self.stuff_area,
                                           self.void_label)

    @torch.no_grad()
    def postprocess(self, sem, instance_cells):
        sem = self._harden_seg(sem)[0]
        return self.get_panoptic_seg(sem, instance_cells)

    def __call__(self, image, size, upsampling=1):
        assert math.log(upsampling, 2).is_integer(), "Upsampling factor not log base 2!"
        assert image.ndim == 4 and image.size(0) == 1
        h, w = size
        image = factor_pad(image, self.padding_factor)
        image = self.to_model_device(image)
        model_out = self.infer(image, int(2 + math.log(upsampling, 2)))
        instance_cells = self.get_instance_cells(model_out['ctr_hmp'], model_out['offsets'], upsampling)
        pan_seg = self.postprocess(model_out['sem'], instance_cells)
        return pan_seg[..., :h, :w]

    @torch.no_grad()
    def postprocess_stack(self, heads, upsampling=1):
        return panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'],
                              coarse_boundaries=self.coarse_boundaries, upsampling=upsampling, **self._stack_params())


class PanopticDeepLabRenderEngine3d(_MedianQueue, PanopticDeepLabRenderEngine):
    """engines.py:327-394"""

    def __init__(self, model, thing_list, label_divisor=1000, stuff_area=64, void_label=0, nms_threshold=0.1,
                 nms_kernel=7, confidence_thr=0.5, median_kernel_size=3, padding_factor=16, coarse_boundaries=True,
                 **kwargs):
        super().__init__(model=model, thing_list=thing_list, label_divisor=label_divisor, stuff_area=stuff_area,
                         void_label=void_label, nms_threshold=nms_threshold, nms_kernel=nms_kernel,
                         confidence_thr=confidence_thr, median_kernel_size=median_kernel_size,
                         padding_factor=padding_factor, coarse_boundaries=coarse_boundaries)

    def end(self, upsampling=1):
        final_segs = []
        for model_out in list(self.median_queue)[self.mid_idx + 1:]:
            h, w = model_out['size']
            cells = self.get_instance_cells(model_out['ctr_hmp'], model_out['offsets'], upsampling)
            final_segs.append(self.postprocess(model_out['sem'], cells)[..., :h, :w])
        return final_segs

    def __call__(self, image, size, upsampling=1):
        assert math.log(upsampling, 2).is_integer(), "Upsampling factor not log base 2!"
        assert image.ndim == 4 and image.size(0) == 1
        h, w = size
        image = factor_pad(image, self.padding_factor)
        image = self.to_model_device(image)
        model_out = self.infer(image, int(2 + math.log(upsampling, 2)))
        model_out['size'] = size
        self.enqueue(model_out)
        median_out = self.get_next(keys=['sem'])
        if median_out is None:
            return None
        cells = self.get_instance_cells(median_out['ctr_hmp'], median_out['offsets'], upsampling)
        pan_seg = self.postprocess(median_out['sem'], cells)
        return pan_seg[..., :h, :w]


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
