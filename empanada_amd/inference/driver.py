"""Volume-level driver of the whole-stack path: what scripts/pdl_inference3d.py:110-233 does slice by slice --
for every plane: model -> engine -> RLE -> forward / backward matching -> trackers -> filters; then the consensus over
the planes (orthoplane mode) or the single plane's trackers (stack mode); then `fill_volume` into a zarr dataset per
class -- as ONE call that keeps the volume, the head tensors of a plane and every run table resident in HBM.

    engine = PanopticDeepLabRenderEngine3d(model, **engine_params)          # any of the four engines
    result = infer_volume(engine, volume_u8, norms=desc['norms'], labels=desc['labels'],
                          class_names=desc['class_names'], out=ZarrV2Group('out.zarr'))

With torch.distributed initialised (one process per GPU, backend 'nccl') every rank calls it with the same volume: the
slices of every plane are split into contiguous blocks over the ranks and each rank writes its own z-slab
(empanada_amd/inference/sharded.py).  Results are identical to the per-slice protocol on the same head tensors
(tests/test_pipeline_gpu.py); bench.py times the same sequence with planted heads.
"""
import numpy as np
import torch

from ..data import DeviceVolume
from . import sharded
from .engines import logits_to_prob

__all__ = ['infer_volume']

_AXES = {'xy': 0, 'xz': 1, 'yz': 2}


@torch.no_grad()
def _plane_heads(engine, dv, axis, lo, hi, batch_pixels, render_steps):
    """model forward over slices [lo, hi) of one plane -> resident {'sem' probabilities, 'ctr_hmp', 'offsets'} at the
    padded size (the reference pads every slice to a multiple of padding_factor and crops the labels, engines.py:351-394)"""
    hp, wp = dv.padded_shape(axis)
    per = max(1, batch_pixels // (hp * wp))
    render = hasattr(engine, 'coarse_boundaries')                      # the PointRend ("render") engines
    outs = {'sem': [], 'ctr_hmp': [], 'offsets': []}
    for s in range(lo, hi, per):
        x = dv.batch(axis, s, min(hi, s + per)).contiguous(memory_format=torch.channels_last)
        if render:
            o = engine.model(x, render_steps, interpolate_ins=not engine.coarse_boundaries)
        else:
            o = engine.model(x)
        outs['sem'].append(logits_to_prob(o['sem_logits']).float())
        outs['ctr_hmp'].append(o['ctr_hmp'].float())
        outs['offsets'].append(o['offsets'].float())
    return {k: torch.cat(v, dim=0).contiguous() for k, v in outs.items()}


def infer_volume(engine, volume, *, norms, labels, axes=('xy', 'xz', 'yz'), merge_iou_thr=0.25, merge_ioa_thr=0.25,
                 min_size=500, min_span=4, pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False, class_names=None,
                 out=None, batch_pixels=32 * 1024 * 1024, render_steps=2, group=None):
    """3D panoptic inference of a (D, H, W) uint8 volume (numpy array, tensor or DeviceVolume) with a 3d engine.

    axes: ('xy',) = stack mode, ('xy', 'xz', 'yz') = orthoplane mode with consensus (pdl_inference3d.py:92-96).
    labels: all class ids; engine.thing_list says which are instance classes.  out: ZarrV2Group (or anything with the
    same create_dataset) -- one dataset '<class name>_pred' per class, uint32 for thing classes and uint8 for stuff,
    chunks (1, Y, X) (pdl_inference3d.py:225-233); rank 0 creates them, every rank writes its slab.
    Returns {'volumes': {class: the rank's (z1 - z0, Y, X) device slab}, 'z_range': (z0, z1),
             'instances': {class: number of instances kept}, 'datasets': {class: array or None}}."""
    rank, world = sharded._world(group)
    labels = list(labels)
    thing_list = list(engine.thing_list)
    div = engine.label_divisor
    factor = int(getattr(engine, 'padding_factor', 16))
    dv = volume if isinstance(volume, DeviceVolume) else DeviceVolume(volume, norms['mean'], norms['std'], factor,
                                                                      next(engine.model.parameters()).device)
    shape3d = dv.shape
    params = dict(thing_list=thing_list, label_divisor=div, stuff_area=engine.stuff_area, void_label=engine.void_label,
                  nms_threshold=engine.nms_threshold, nms_kernel=engine.nms_kernel,
                  confidence_thr=engine.confidence_thr, median_kernel_size=getattr(engine, 'ks', 1),
                  coarse_boundaries=bool(getattr(engine, 'coarse_boundaries', False)))
    planes, base = {}, 0
    for axis in axes:
        n = dv.n_slices(axis)
        b = sharded.shard_bounds(n, world)
        lo, hi = int(b[rank]), int(b[rank + 1])
        h, w = dv.plane_shape(axis)
        heads = _plane_heads(engine, dv, axis, lo, hi, batch_pixels, render_steps)
        pan = sharded.sharded_panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'], group=group, **params)
        del heads
        pan = pan[:, :h, :w].contiguous()
        planes[axis] = sharded.track_plane(pan, axis, shape3d, labels, thing_list, div, merge_iou_thr, merge_ioa_thr,
                                           inst_base=base, group=group)
        base += planes[axis].n_inst
        del pan
    if len(axes) == 1:
        # stack mode: the plane's own trackers are the result (pdl_inference3d.py:222-223)
        assert axes[0] == 'xy', "stack mode runs along z (axes=('xy',))"
        vols, (z0, z1), counts = sharded.plane_volume(planes['xy'], labels, thing_list, min_size, min_span, group=group)
    else:
        cons, vols, (z0, z1) = sharded.consensus_volume(planes, shape3d, labels, thing_list, pixel_vote_thr,
                                                        cluster_iou_thr, bypass, min_size, min_span, group=group)
        counts = {c: int(cons[c].alive.sum()) for c in labels}
    datasets = {c: None for c in labels}
    if out is not None:
        names = {c: f"{(class_names or {}).get(c, c)}_pred" for c in labels}
        if rank == 0:
            for c in labels:
                out.create_dataset(names[c], shape=shape3d, dtype=np.uint32 if c in thing_list else np.uint8,
                                   overwrite=True, chunks=(1, None, None))
        if world > 1:
            torch.distributed.barrier(group=group)
        for c in labels:
            datasets[c] = out[names[c]]
            host = vols[c].view(torch.int32).cpu().numpy().view(np.uint32) if c in thing_list else vols[c].cpu().numpy()
            datasets[c].write_slab(z0, host)
    return {'volumes': vols, 'z_range': (z0, z1), 'instances': counts, 'datasets': datasets}
