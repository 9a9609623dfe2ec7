"""GPU: the volume-level driver (empanada_amd/inference/driver.py: model forward on every plane -> engines' whole-stack
post-processing -> device trackers -> consensus / stack volume -> zarr) against a manual composition of the building
blocks the other tests pin to the reference (panoptic_stack, track_stack, filters, create_*_consensus,
fill_volume_device), with a real model forward in between: plane views of the resident volume, padding to the engine's
padding factor and cropping, class handling, per-class datasets."""
import numpy as np
import pytest
import torch

from empanada_amd import synthetic as SY

pytestmark = pytest.mark.gpu

NORMS = dict(mean=0.508979, std=0.148561)


def _engine(ks=3, render=False):
    from empanada_amd.inference import engines as EN
    from empanada_amd.models import PanopticDeepLab, PanopticDeepLabPR, prepare_for_inference, synthesize_weights
    from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
    cls = PanopticDeepLabPR if render else PanopticDeepLab
    model = synthesize_weights(cls(encoder='resnet18', num_classes=3))
    with torch.no_grad():                                  # offsets of a few pixels instead of hundreds
        model.ins_xy.head[1].weight.mul_(2e-2)
    model = prepare_for_inference(model, 'cuda')
    for m in model.modules():                              # the hand-written kernels: run-to-run identical outputs
        if isinstance(m, FusedConvBNAct) and 'direct' in m.candidates(False):
            m.impl = 'direct'
    kw = dict(thing_list=[1], label_divisor=1000, stuff_area=16, void_label=0, nms_threshold=0.1, nms_kernel=7,
              confidence_thr=0.5, median_kernel_size=ks, padding_factor=32)
    if render:
        return EN.PanopticDeepLabRenderEngine3d(model, coarse_boundaries=True, **kw)
    return EN.PanopticDeepLabEngine3d(model, **kw)


def _manual(engine, vol, axes, min_size, min_span):
    from empanada_amd.data import DeviceVolume
    from empanada_amd.inference import driver, filters
    from empanada_amd.inference import patterns as PA
    from empanada_amd.inference.postprocess import panoptic_stack
    dv = DeviceVolume(vol, NORMS['mean'], NORMS['std'], int(getattr(engine, 'padding_factor', 16)), 'cuda')
    labels, thing = [1, 2], [1]
    trackers = {}
    for axis in axes:
        h, w = dv.plane_shape(axis)
        heads = driver._plane_heads(engine, dv, axis, 0, dv.n_slices(axis), 1 << 22, 2)
        pan, emitted = panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'], thing_list=thing,
                                      label_divisor=1000, stuff_area=16, void_label=0, nms_threshold=0.1, nms_kernel=7,
                                      confidence_thr=0.5, median_kernel_size=engine.ks,
                                      coarse_boundaries=bool(getattr(engine, 'coarse_boundaries', False)))
        assert len(emitted) == dv.n_slices(axis)
        trackers[axis] = PA.track_stack(pan[:, :h, :w].contiguous(), axis, vol.shape, labels, thing, 1000, 0.25, 0.25)
        for tr in trackers[axis]:
            filters.remove_small_objects(tr, min_size)
            filters.remove_pancakes(tr, min_span)
    out = {}
    for c in labels:
        cts = PA.get_axis_trackers_by_class(trackers, c)
        if len(axes) == 1:
            con = cts[0]
            v = PA.fill_volume_device(vol.shape, [con]).cpu().numpy().astype(np.uint32)
            out[c] = v if c in thing else (v > 0).astype(np.uint8)
        elif c in thing:
            con = PA.create_instance_consensus(cts, 2, 0.75, False)
            filters.remove_small_objects(con, min_size)
            filters.remove_pancakes(con, min_span)
            out[c] = PA.fill_volume_device(vol.shape, [con]).cpu().numpy().astype(np.uint32)
        else:
            con = PA.create_semantic_consensus(cts, 2)
            out[c] = PA.fill_volume_device(vol.shape, [con], dtype=torch.uint8).cpu().numpy()
    return out


@pytest.mark.parametrize('render', [False, True])
@pytest.mark.parametrize('axes', [('xy', 'xz', 'yz'), ('xy',)])
def test_infer_volume_equals_manual_composition(tmp_path, axes, render):
    from empanada_amd.inference.driver import infer_volume
    from empanada_amd.zarr_utils import ZarrV2Group, open_zarr
    vol = SY.em_volume((40, 72, 88), seed=3)               # 72 and 88 are not multiples of the padding factor 32
    engine = _engine(ks=3, render=render)
    out = ZarrV2Group(str(tmp_path / 'pred.zarr'))
    res = infer_volume(engine, vol, norms=NORMS, labels=[1, 2], axes=axes, min_size=30, min_span=2,
                       class_names={1: 'mito', 2: 'er'}, out=out, batch_pixels=1 << 22)
    exp = _manual(engine, vol, axes, 30, 2)
    assert res['z_range'] == (0, 40)
    for c, name, dt in ((1, 'mito_pred', np.uint32), (2, 'er_pred', np.uint8)):
        got = res['volumes'][c]
        got = got.view(torch.int32).cpu().numpy().view(np.uint32) if c == 1 else got.cpu().numpy()
        np.testing.assert_array_equal(got, exp[c], err_msg=f'class {c}')
        arr = open_zarr(str(tmp_path / 'pred.zarr' / name))
        assert arr.dtype == dt and tuple(arr.chunks) == (1, 72, 88)
        np.testing.assert_array_equal(arr[...], exp[c])
    assert res['instances'][1] == len(np.unique(exp[1])) - 1
    assert exp[1].max() > 0 or exp[2].max() > 0, "the random model should segment something"
