"""GPU: the product package end to end (engines -> RLE -> matching -> trackers -> consensus -> fill),
through both protocols (per-slice drop-in API and whole-stack fast path), against the fixtures the
reference produced (tests/golden/engines.npz, pipeline.npz, consensus_kat.npz, matcher_kat.npz)."""
import numpy as np
import pytest
import torch

from conftest import assert_instances_equal, dense_tol, load_golden, unpack_instances
from empanada_amd import synthetic as SY

pytestmark = pytest.mark.gpu


class Stub(torch.nn.Module):
    """hands out pre-computed head tensors slice by slice; 'sem_logits' already holds probabilities"""

    def __init__(self, heads):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.heads, self.t = heads, 0

    def forward(self, x, *a, **k):
        o = {k2: v[self.t:self.t + 1].clone().to(self.p.device) for k2, v in self.heads.items()}
        o['sem_logits'] = o.pop('sem')
        self.t += 1
        return o


@pytest.fixture()
def prob_passthrough(monkeypatch):
    from empanada_amd.inference import engines
    monkeypatch.setattr(engines, 'logits_to_prob', lambda x: x)
    return engines


def _engine_case(g, i):
    ks, coarse, render, C, nk = (int(x) for x in g[f'e{i}_par'])
    heads = {'sem': torch.from_numpy(g[f'e{i}_sem']), 'ctr_hmp': torch.from_numpy(g[f'e{i}_ctr']),
             'offsets': torch.from_numpy(g[f'e{i}_off'])}
    kw = dict(thing_list=[int(t) for t in g[f'e{i}_thing']], label_divisor=1000, stuff_area=32, void_label=0,
              nms_threshold=0.1, nms_kernel=nk, confidence_thr=float(g[f'e{i}_thr']), median_kernel_size=ks)
    return heads, kw, bool(coarse), bool(render), g[f'e{i}_pan']


def test_engines_per_slice_protocol(prob_passthrough):
    EN = prob_passthrough
    g = load_golden('engines')
    for i in range(int(g['n'])):
        heads, kw, coarse, render, exp = _engine_case(g, i)
        S, _, H, W = heads['sem'].shape
        stub = Stub(heads).cuda()
        outs = []
        if render:
            eng = EN.PanopticDeepLabRenderEngine3d(stub, padding_factor=16, coarse_boundaries=coarse, **kw)
            for t in range(S):
                o = eng(torch.zeros(1, 1, H, W), (H - 3, W - 5))
                if o is not None:
                    outs.append(o.cpu().numpy())
            outs += [o.cpu().numpy() for o in eng.end()]
        else:
            eng = EN.PanopticDeepLabEngine3d(stub, **kw)
            for t in range(S):
                o = eng(torch.zeros(1, 1, H, W))
                if o is not None:
                    outs.append(o.cpu().numpy())
            outs += [o.cpu().numpy() for o in eng.end()]
        got = np.stack(outs)
        assert got.dtype == np.int64
        np.testing.assert_array_equal(got, exp, err_msg=f'engine case {i}')


def test_engines_whole_stack_protocol():
    from empanada_amd.inference.postprocess import panoptic_stack
    g = load_golden('engines')
    for i in range(int(g['n'])):
        heads, kw, coarse, render, exp = _engine_case(g, i)
        S, _, H, W = heads['sem'].shape
        pan, emitted = panoptic_stack(heads['sem'].cuda(), heads['ctr_hmp'].cuda(), heads['offsets'].cuda(),
                                      coarse_boundaries=coarse if render else False, **kw)
        assert emitted == list(range(S))
        got = pan.cpu().numpy().astype(np.int64)
        exp = exp.reshape(S, *exp.shape[-2:])
        np.testing.assert_array_equal(got[:, :exp.shape[1], :exp.shape[2]], exp, err_msg=f'engine case {i}')


def test_short_stack_loses_the_slices_the_reference_loses():
    from empanada_amd.inference.postprocess import panoptic_stack
    from oracle import postprocess as OP
    lab, cls = SY.planted_labels((4, 48, 48), fill=0.2, rmin=4, rmax=8, seed=3)
    heads = SY.planted_heads(lab, cls, 'xy', seed=1)
    kw = dict(thing_list=[1], label_divisor=1000, stuff_area=32, void_label=0, nms_threshold=0.1, nms_kernel=7,
              confidence_thr=0.5, median_kernel_size=7)
    pan, emitted = panoptic_stack(heads['sem'].cuda(), heads['ctr_hmp'].cuda(), heads['offsets'].cuda(),
                                  coarse_boundaries=False, **kw)
    sem, ctr, off = (heads[k].numpy() for k in ('sem', 'ctr_hmp', 'offsets'))
    exp = OP.engine3d_stack([sem[t:t + 1] for t in range(4)], [ctr[t:t + 1] for t in range(4)],
                            [off[t:t + 1] for t in range(4)], coarse_boundaries=False, render=True, **kw)
    assert emitted == [0, 1, 2]          # ks=7: slices 0..2 raw, slice 3 (the middle slot) is lost
    np.testing.assert_array_equal(pan.cpu().numpy().astype(np.int64), np.stack(exp)[:, 0])


def test_matcher_kat():
    """reference tests/test_matcher.py:54-66 through the product RLEMatcher"""
    from empanada_amd.inference import matcher, rle
    g = load_golden('matcher_kat')
    m = matcher.RLEMatcher(1, 1000, 0.25, 0.25, True)
    t = rle.pan_seg_to_rle_seg(g['target'], [1], 1000, [1], False)
    r = rle.pan_seg_to_rle_seg(g['match'], [1], 1000, [1], False)
    m.initialize_target(t[1])
    r[1] = m(r[1], update_target=False)
    np.testing.assert_array_equal(rle.rle_seg_to_pan_seg(r, (200, 200)), g['out'])


def _params(g, i):
    C, ks, _, head_seed = (int(x) for x in g[f'p{i}_par'])
    thing = [1] if C == 1 else list(range(1, C))
    labels = [1] if C == 1 else list(range(1, C + 1))
    return C, ks, head_seed, thing, labels


def test_reference_protocol_matching_and_tracking():
    """pan_seg_to_rle_seg + apply_matchers + backward_matching + update_trackers, slice by slice"""
    from empanada_amd.inference import patterns as PA
    from empanada_amd.inference import rle
    g = load_golden('pipeline')
    for i in range(int(g['n'])):
        C, ks, head_seed, thing, labels = _params(g, i)
        shape = g[f'p{i}_lab'].shape
        trackers = PA.create_axis_trackers({'xy': 0, 'xz': 1, 'yz': 2}, labels, 1000, shape)
        for name in ('xy', 'xz', 'yz'):
            pans = g[f'p{i}_{name}_pan'].astype(np.int64)
            matchers = PA.create_matchers(thing, 1000, 0.25, 0.25)
            stack = []
            for pan in pans:
                stack.append(PA.apply_matchers(rle.pan_seg_to_rle_seg(pan, labels, 1000, thing, True), matchers))
            fwd = np.stack([rle.rle_seg_to_pan_seg(rs, pans[0].shape) for rs in stack])
            np.testing.assert_array_equal(fwd, g[f'p{i}_{name}_fwd'])
            for idx, rs in PA.backward_matching(stack, matchers, len(pans)):
                PA.update_trackers(rs, idx, trackers[name])
            bwd = np.stack([rle.rle_seg_to_pan_seg(rs, pans[0].shape) for rs in stack])
            np.testing.assert_array_equal(bwd, g[f'p{i}_{name}_bwd'])
            PA.finish_tracking(trackers[name])
            for tr in trackers[name]:
                assert_instances_equal(tr.instances, unpack_instances(g, f'p{i}_{name}_tr{tr.class_id}'))


def run_fast_pipeline(lab, cls, C, ks, head_seed, thing, labels, min_size=100, min_span=3):
    from empanada_amd.inference import filters
    from empanada_amd.inference import patterns as PA
    from empanada_amd.inference.postprocess import panoptic_stack
    shape = lab.shape
    trackers, raw, pans = {}, {}, {}
    for name in ('xy', 'xz', 'yz'):
        heads = SY.planted_heads(lab, cls, name, n_classes=C, seed=head_seed, coarse=False)
        pan, emitted = panoptic_stack(heads['sem'].cuda(), heads['ctr_hmp'].cuda(), heads['offsets'].cuda(),
                                      thing_list=thing, label_divisor=1000, stuff_area=16, void_label=0,
                                      nms_threshold=0.1, nms_kernel=7, confidence_thr=0.5, median_kernel_size=ks,
                                      coarse_boundaries=False)
        pans[name] = pan
        trackers[name] = PA.track_stack(pan, name, shape, labels, thing, 1000, 0.25, 0.25)
        raw[name] = {t.class_id: dict(t.instances) for t in trackers[name]}
        for t in trackers[name]:
            filters.remove_small_objects(t, min_size=min_size)
            filters.remove_pancakes(t, min_span=min_span)
    cons = {}
    for cid in labels:
        cts = PA.get_axis_trackers_by_class(trackers, cid)
        if cid in thing:
            con = PA.create_instance_consensus(cts, 2, 0.75, False)
            filters.remove_small_objects(con, min_size=min_size)
            filters.remove_pancakes(con, min_span=min_span)
        else:
            con = PA.create_semantic_consensus(cts, 2)
        cons[cid] = con
    return pans, raw, cons


def test_whole_stack_pipeline_matches_reference():
    from empanada_amd.inference import patterns as PA
    g = load_golden('pipeline')
    for i in range(int(g['n'])):
        C, ks, head_seed, thing, labels = _params(g, i)
        lab, cls = g[f'p{i}_lab'], g[f'p{i}_cls']
        pans, raw, cons = run_fast_pipeline(lab, cls, C, ks, head_seed, thing, labels)
        for name in ('xy', 'xz', 'yz'):
            np.testing.assert_array_equal(pans[name].cpu().numpy().astype(np.int64), g[f'p{i}_{name}_pan'])
            for cid in labels:
                assert_instances_equal(raw[name][cid], unpack_instances(g, f'p{i}_{name}_tr{cid}'))
        for cid in labels:
            assert_instances_equal(cons[cid].instances, unpack_instances(g, f'p{i}_con{cid}'))
            vol = PA.fill_volume_device(lab.shape, [cons[cid]]).cpu().numpy()
            np.testing.assert_array_equal(vol, g[f'p{i}_vol{cid}'])
            host = np.zeros(lab.shape, dtype=np.uint32)
            PA.fill_volume(host, cons[cid].instances)
            np.testing.assert_array_equal(host, g[f'p{i}_vol{cid}'])
        # the device-resident route bench.py takes: PlaneTracks -> consensus on tables -> paint, nothing on the host
        dcons, dvols = run_device_pipeline(pans, lab.shape, labels, thing)
        for cid in labels:
            assert_instances_equal(dcons[cid].instances(), unpack_instances(g, f'p{i}_con{cid}'))
            np.testing.assert_array_equal(dvols[cid].cpu().numpy().astype(np.uint32), g[f'p{i}_vol{cid}'])


def run_device_pipeline(pans, shape, labels, thing, min_size=100, min_span=3, div=1000):
    from empanada_amd.inference import sharded
    planes, base = {}, 0
    for name in ('xy', 'xz', 'yz'):
        planes[name] = sharded.track_plane(pans[name], name, shape, labels, thing, div, 0.25, 0.25, inst_base=base)
        base += planes[name].n_inst
    cons, vols, (z0, z1) = sharded.consensus_volume(planes, shape, labels, thing, 2, 0.75, False, min_size, min_span)
    assert (z0, z1) == (0, shape[0])
    return cons, vols


def test_consensus_kats():
    """reference tests/test_consensus.py cases through the product consensus (HIP voting / intersections)"""
    from empanada_amd import consensus as CO
    from empanada_amd.array_utils import numpy_fill_instances
    from empanada_amd.inference import rle, tracker
    g = load_golden('consensus_kat')
    vols = g['vols']
    shape = vols[0].shape
    trs = [tracker.InstanceTracker(1, 1000, shape, axis='xy') for _ in range(3)]
    for v, tr in zip(vols, trs):
        segs, _ = rle.stack_to_rle_segs(torch.from_numpy(v.astype(np.int32)).cuda().view(torch.uint32), [1], 1000,
                                        [1], force_connected=False)
        for z in range(shape[0]):
            tr.update(segs[z][1], z)
        tr.finish()
    for j in range(6):
        vote, iou_thr, bypass = g[f'k{j}_par']
        inst = CO.merge_objects_from_trackers(trs, int(vote), float(iou_thr), bool(bypass))
        assert_instances_equal(inst, unpack_instances(g, f'k{j}_inst'))
        vol = numpy_fill_instances(np.zeros(shape, np.uint32), inst).ravel()
        np.testing.assert_array_equal(vol, np.repeat(g[f'k{j}_val'], g[f'k{j}_ln']))
    for j in range(2):
        strs = []
        for v in vols:
            tr = tracker.InstanceTracker(1, 1000, shape, axis='xy')
            sem = ((v > 0).astype(np.int32) * 1000)
            segs, _ = rle.stack_to_rle_segs(torch.from_numpy(sem).cuda().view(torch.uint32), [1], 1000, [], False)
            for z in range(shape[0]):
                tr.update(segs[z][1], z)
            tr.finish()
            strs.append(tr)
        inst = CO.merge_semantic_from_trackers(strs, int(g[f's{j}_vote']))
        assert_instances_equal(inst, unpack_instances(g, f's{j}_inst'))


def test_trackers_and_chunked_fill():
    from empanada_amd.inference import rle, tracker
    from empanada_amd.zarr_utils import ChunkedArray, zarr_fill_instances
    g = load_golden('trackers')
    vol = g['vol']
    for name, ax in (('xy', 0), ('xz', 1), ('yz', 2)):
        tr = tracker.InstanceTracker(1, 1000, vol.shape, axis=name)
        stack = np.ascontiguousarray(np.moveaxis(vol, ax, 0)).astype(np.int32)
        segs, _ = rle.stack_to_rle_segs(torch.from_numpy(stack).cuda().view(torch.uint32), [1], 1000, [1], False)
        for idx in range(vol.shape[ax]):
            tr.update(segs[idx][1], idx)
        tr.finish()
        assert_instances_equal(tr.instances, unpack_instances(g, f't_{name}'))
        if name == 'xy':
            # reference tests/test_tracking.py:61-71 with arbitrary chunk shapes; expected volumes come from the
            # reference's own zarr_fill_instances (incl. its handling of runs that wrap over a narrow last chunk)
            for j in range(int(g['zf_n'])):
                chunks = tuple(int(c) for c in g[f'zf{j}_chunks'])
                arr = ChunkedArray(np.zeros(vol.shape, np.uint32), chunks)
                zarr_fill_instances(arr, tr.instances, 4)
                np.testing.assert_array_equal(arr.array, g[f'zf{j}_vol'], err_msg=str(chunks))
                if j % 4 == 0:               # the same fill into an on-disk zarr v2 array (scripts/pdl_inference3d.py:228-233)
                    import tempfile
                    from empanada_amd.zarr_utils import ZarrV2Group, open_zarr
                    with tempfile.TemporaryDirectory() as d:
                        ds = ZarrV2Group(d + '/out.zarr').create_dataset('mito_pred', shape=vol.shape, dtype=np.uint32,
                                                                         overwrite=True, chunks=chunks)
                        zarr_fill_instances(ds, tr.instances, 4)
                        np.testing.assert_array_equal(open_zarr(ds.path)[...], g[f'zf{j}_vol'], err_msg=str(chunks))


def test_model_forward_matches_cpu_within_tolerance():
    """D1: fp32 forward on the GPU (MIOpen) vs the same module on the host; tolerance from SURVEY 8(c)."""
    from empanada_amd.models import PanopticDeepLab, prepare_for_inference, synthesize_weights
    torch.manual_seed(0)
    m = synthesize_weights(PanopticDeepLab(encoder='resnet50', num_classes=1)).eval()
    # damp the synthetic head scale (He weights are hot): logits come out O(0.05), so the bound below is RELATIVE to
    # the head's own |ref|_inf -- 1e-4 of full scale, no absolute floor a sloppy kernel could hide under
    with torch.no_grad():
        for head in (m.semantic_head, m.ins_center, m.ins_xy):
            head.head[1].weight.mul_(1e-3)
    x = torch.randn(2, 1, 128, 128)
    with torch.no_grad():
        ref = m(x)
        g = prepare_for_inference(m, 'cuda')
        out = g(x.cuda().contiguous(memory_format=torch.channels_last))
    for k in ref:
        scale = float(ref[k].abs().max())
        err = float((out[k].float().cpu() - ref[k]).abs().max())
        assert err <= dense_tol(scale), (k, err, scale)


@pytest.mark.parametrize('force', ['direct', 'wino', 'wino_sep', 'wino4', 'wino3', 'tuned'])
def test_model_forward_with_hip_convolutions(force):
    """D4 / D5 inside the model: every conv + BN call site forced onto the implicit-GEMM kernel, onto Winograd
    where it applies, or left to the tuner -- same tolerance against the host module as the MIOpen path."""
    import copy
    from empanada_amd.models import PanopticDeepLab, prepare_for_inference, synthesize_weights, tune_fused_convs
    from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
    torch.manual_seed(0)
    m = synthesize_weights(PanopticDeepLab(encoder='resnet50', num_classes=1)).eval()
    with torch.no_grad():
        for head in (m.semantic_head, m.ins_center, m.ins_xy):
            head.head[1].weight.mul_(1e-3)
    x = torch.randn(2, 1, 128, 128)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = m(x)
        g = prepare_for_inference(copy.deepcopy(m), 'cuda')
        sites = [mod for mod in g.modules() if isinstance(mod, FusedConvBNAct)]
        assert len(sites) > 50
        if force == 'tuned':
            rep = tune_fused_convs(g, xd, reps=2)
            assert len(rep) == len(sites) and all(best in t for best, t in rep.values())
        else:
            g(xd)                                        # records the call-site shapes
            n = 0
            for mod in sites:
                if force in mod.candidates(mod._seen[1]):
                    mod.impl = force
                    n += 1
            assert n >= (5 if force.startswith('wino') else 50)
        out = g(xd)
    for k in ref:
        scale = float(ref[k].abs().max())
        err = float((out[k].float().cpu() - ref[k]).abs().max())
        assert err <= dense_tol(scale), (force, k, err, scale)


@pytest.mark.parametrize('force', ['hand_written', 'tuned'])
def test_regnety_forward_with_hip_convolutions(force):
    """PanopticBiFPN / RegNetY-6.4GF (encoders/regnet.py, blocks.py:35-50): the grouped 3x3 convolutions on
    emp_gconv3x3_bn_act_nhwc, the 1x1 convolutions with Cin = 144 / 1296 on the 16-wide K-slab variant, the per-pixel
    squeeze-excite as two fused launches (gate epilogue) and shortcut + ReLU in the last convolution's epilogue --
    same tolerance against the host module as the MIOpen path.  'tuned': whatever the tuner picks per site."""
    import copy
    from empanada_amd.models import PanopticBiFPN, prepare_for_inference, synthesize_weights, tune_fused_convs
    from empanada_amd.models.panoptic_bifpn import FusedSqueezeExcite, _RegBlock
    from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
    torch.manual_seed(0)
    m = synthesize_weights(PanopticBiFPN(encoder='regnety_6p4gf', num_classes=1)).eval()
    with torch.no_grad():
        for head in (m.semantic_head, m.ins_center, m.ins_xy):
            head.head[1].weight.mul_(1e-3)
    x = torch.randn(2, 1, 128, 128)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = m(x)
        g = prepare_for_inference(copy.deepcopy(m), 'cuda')
        blocks = [mod for mod in g.modules() if isinstance(mod, _RegBlock)]
        assert len(blocks) == 25 and all(b.fused_tail and isinstance(b.bottleneck.se, FusedSqueezeExcite) for b in blocks)
        sites = [mod for mod in g.modules() if isinstance(mod, FusedConvBNAct)]
        if force == 'tuned':
            rep = tune_fused_convs(g, xd, reps=2)
            assert len(rep) == len(sites)
        else:
            g(xd)                                        # records the call-site shapes
            n = {'grouped': 0, 'direct': 0}
            for mod in sites:
                cand = mod.candidates(mod._seen[1])
                for impl in ('grouped', 'direct'):
                    if impl in cand:
                        mod.impl = impl
                        n[impl] += 1
            assert n['grouped'] == 25 and n['direct'] >= 50, n
        out = g(xd)
    for k in ref:
        scale = float(ref[k].abs().max())
        err = float((out[k].float().cpu() - ref[k]).abs().max())
        assert err <= dense_tol(scale), (force, k, err, scale)


def test_graphed_forward_replays_the_model():
    """models/graphed.py: the batch-1 forward captured as a HIP graph (hand-written kernels + MIOpen in one capture)
    returns what the eager forward returns, for fresh inputs, a second shape, and outputs that stay valid after later
    replays (the engines keep them in the median queue).  Two eager runs of the same input already differ in the last
    bit at batch 1 (MIOpen picks split-K kernels with atomic accumulation for the stem), so the comparison allows
    1e-6 * max(1, |x|_inf); a stale or missing node in the graph would be off by orders of magnitude."""
    from empanada_amd.models import GraphedForward, PanopticDeepLab, prepare_for_inference, synthesize_weights
    from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
    torch.manual_seed(3)
    m = synthesize_weights(PanopticDeepLab(encoder='resnet50', num_classes=1)).eval()
    with torch.no_grad():
        for head in (m.semantic_head, m.ins_center, m.ins_xy):
            head.head[1].weight.mul_(1e-3)
        net = prepare_for_inference(m, 'cuda')
        xs = [torch.randn(1, 1, 256, 256, device='cuda').contiguous(memory_format=torch.channels_last) for _ in range(3)]
        xs.append(torch.randn(2, 1, 128, 192, device='cuda').contiguous(memory_format=torch.channels_last))
        net(xs[0])
        n = 0
        for mod in net.modules():                      # a mix of implementations inside one capture
            if isinstance(mod, FusedConvBNAct) and mod._seen is not None:
                cand = mod.candidates(mod._seen[1])
                mod.impl = ('wino4' if 'wino4' in cand else 'direct' if 'direct' in cand else 'miopen') if n % 3 else 'miopen'
                n += 1
        eager = [{k: v.clone() for k, v in net(x).items()} for x in xs]
        graphed = GraphedForward(net)
        outs = [graphed(x) for x in xs] + [graphed(xs[0])]
        assert len(graphed._graphs) == 2
        for got, exp in zip(outs, eager + [eager[0]]):
            for k in exp:
                err = float((got[k] - exp[k]).abs().max())
                assert err <= 1e-6 * max(1.0, float(exp[k].abs().max())), (k, err)
        assert next(graphed.parameters()).is_cuda


def test_graphed_pointrend_replays_survive_other_launches():
    """Regression for the GPU memory fault of round 3: a captured PointRend forward must give
    bit-identical replays however many unrelated kernels run between them.  A `hipMemsetAsync` inside the captured step
    (a memset node) broke exactly this -- wrong point indices after ~2 000 launches, a memory fault a few thousand
    later -- so everything capturable zeroes with kernels.  4 images (multi-image top-k), two shapes, 20 000 launches."""
    from empanada_amd import _hip
    from empanada_amd.models import GraphedForward, PanopticDeepLabPR, prepare_for_inference, synthesize_weights
    mito = dict(encoder='resnet50', num_classes=1, stage4_stride=16, decoder_channels=256, low_level_stages=[1],
                low_level_channels_project=[32], atrous_rates=[2, 4, 6], aspp_channels=None, aspp_dropout=0.5,
                ins_decoder=True, ins_ratio=0.5)
    torch.manual_seed(5)
    m = synthesize_weights(PanopticDeepLabPR(**mito)).eval()
    with torch.no_grad():
        for head in (m.semantic_head, m.ins_center, m.ins_xy):
            head.head[1].weight.mul_(1e-3)
        net = prepare_for_inference(m, 'cuda')
        x = torch.randn(4, 1, 256, 256, device='cuda').contiguous(memory_format=torch.channels_last)
        net(x, 2, False)
        # every convolution on this package's (deterministic) kernels: at small batches MIOpen picks split-K kernels
        # that accumulate with atomics, and a last-bit difference upstream reshuffles PointRend's top-k
        from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
        for mod in net.modules():
            if isinstance(mod, FusedConvBNAct) and mod._seen is not None:
                assert 'direct' in mod.candidates(mod._seen[1])
                mod.impl = 'direct'
        eager = {k: v.clone() for k, v in net(x, 2, False).items()}
        graphed = GraphedForward(net, clone_outputs=False)
        first = {k: v.clone() for k, v in graphed(x, 2, False).items()}
        for k in eager:
            assert torch.equal(first[k], eager[k]), k
        xs = torch.randn(1, 32, 4, 4, device='cuda').contiguous(memory_format=torch.channels_last)
        ws = torch.randn(32, 1, 1, 32, device='cuda')
        t = torch.zeros(64, device='cuda')
        for rnd in range(2):
            for _ in range(5000):
                _hip.conv_bn_act_nhwc(xs, ws)
                t.add_(1.0)
            net(x, 2, False)                                         # and an eager forward of the same model
            again = graphed(x, 2, False)
            for k in first:
                assert torch.equal(again[k], first[k]), (rnd, k, int((again[k] != first[k]).sum()))


@pytest.mark.parametrize('force', ['direct', 'wino4'])
def test_model_forward_large_batch_equals_small_batches(force):
    """128 slices of 512^2 in one model call (activations of 2^29 elements = 2 GiB and more: 32-bit byte offsets
    would wrap) against the same slices in calls of 16.  The K-slab and tile choices depend on the grid size, so
    the comparison is to rounding, not bit for bit."""
    from empanada_amd.models import PanopticDeepLab, prepare_for_inference, synthesize_weights
    from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
    torch.manual_seed(1)
    m = synthesize_weights(PanopticDeepLab(encoder='resnet50', num_classes=1)).eval()
    with torch.no_grad():
        for head in (m.semantic_head, m.ins_center, m.ins_xy):
            head.head[1].weight.mul_(1e-3)
        g = prepare_for_inference(m, 'cuda')
        x = torch.randn(128, 1, 512, 512, device='cuda').contiguous(memory_format=torch.channels_last)
        g(x[:16])
        n = 0
        for mod in g.modules():
            if isinstance(mod, FusedConvBNAct) and mod._seen is not None:
                cand = mod.candidates(mod._seen[1])
                mod.impl = force if force in cand else ('direct' if 'direct' in cand else 'miopen')
                n += mod.impl != 'miopen'
        assert n > 50
        big = {k: v.float().cpu() for k, v in g(x).items()}
        for s in range(0, 128, 16):
            small = g(x[s:s + 16])
            for k, v in small.items():
                ref = v.float().cpu()
                scale = float(ref.abs().max())
                err = float((big[k][s:s + 16] - ref).abs().max())
                assert err <= dense_tol(scale), (force, k, s, err, scale)


@pytest.mark.parametrize('encoder,force', [('resnet50', 'direct'), ('resnet50', 'tuned'), ('regnety_6p4gf', 'direct')])
def test_bifpn_forward_with_hip_convolutions(encoder, force):
    """D1, second model family (models/panoptic_bifpn.py:98-108): PanopticBiFPN through the same graph rewrite
    (fused conv + BN, depthwise, up-sampling kernels) against the host module, same tolerance as PanopticDeepLab."""
    import copy
    from empanada_amd.models import PanopticBiFPN, prepare_for_inference, synthesize_weights, tune_fused_convs
    from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
    torch.manual_seed(0)
    m = synthesize_weights(PanopticBiFPN(encoder=encoder, num_classes=1)).eval()
    x = torch.randn(2, 1, 128, 128)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = m(x)
        g = prepare_for_inference(copy.deepcopy(m), 'cuda')
        sites = [mod for mod in g.modules() if isinstance(mod, FusedConvBNAct)]
        assert len(sites) > 20
        if force == 'tuned':
            rep = tune_fused_convs(g, xd, reps=2)
            assert len(rep) > 20
        else:
            g(xd)
            n = 0
            for mod in sites:
                if mod._seen is not None and force in mod.candidates(mod._seen[1]):
                    mod.impl = force
                    n += 1
            assert n > 20
        out = g(xd)
    for k in ref:
        scale = float(ref[k].abs().max())
        err = float((out[k].float().cpu() - ref[k]).abs().max())
        assert err <= dense_tol(scale), (encoder, force, k, err, scale)


def test_sharded_path_world1_equals_tracker_path():
    """bench.py's path (sharded.py with one rank: tables -> chain -> filters on tables -> fill from the run
    table) paints exactly the volume of track_stack -> filters -> fill_volume_device."""
    from empanada_amd.inference import filters, sharded
    from empanada_amd.inference import patterns as PA
    from empanada_amd.inference.postprocess import panoptic_stack
    lab, cls = SY.planted_labels((24, 96, 128), fill=0.2, rmin=4, rmax=12, seed=8)
    heads = SY.planted_heads(lab, cls, 'xy', seed=4)
    kw = dict(thing_list=[1], label_divisor=20000, stuff_area=64, void_label=0, nms_threshold=0.1, nms_kernel=7,
              confidence_thr=0.3, median_kernel_size=7)
    sem, ctr, off = heads['sem'].cuda(), heads['ctr_hmp'].cuda(), heads['offsets'].cuda()
    pan, _ = panoptic_stack(sem, ctr, off, coarse_boundaries=False, **kw)
    pan2 = sharded.sharded_panoptic_stack(sem, ctr, off, coarse_boundaries=False, **kw)
    np.testing.assert_array_equal(pan.cpu().numpy(), pan2.cpu().numpy())
    trs = PA.track_stack(pan, 'xy', lab.shape, [1], [1], 20000, 0.25, 0.25)
    for tr in trs:
        filters.remove_small_objects(tr, 300)
        filters.remove_pancakes(tr, 4)
    exp = PA.fill_volume_device(lab.shape, trs).cpu().numpy()
    got = sharded.sharded_stack_volume(pan2, [1], [1], 20000, 0.25, 0.25, min_size=300, min_span=4).cpu().numpy()
    assert exp.max() > 0 and len(np.unique(exp)) > 5
    np.testing.assert_array_equal(got, exp)


def test_block_lifted_runs_equal_whole_axis():
    """slice-sharded orthoplane tracking, all three axes, without the collectives: two 'virtual ranks' build their
    tables (with halo), the tables are merged and chained as every rank does, each block lifts the runs of its own
    slices with the instance indices of the whole axis, and the concatenation (sorted; yz pieces split at the block
    border joined again) is exactly the run set of track_stack over the whole axis (which the reference fixtures pin)."""
    from empanada_amd.inference import device_tracks as DT
    from empanada_amd.inference import patterns as PA
    from empanada_amd.inference import sharded
    from empanada_amd.inference.postprocess import panoptic_stack
    shape = (30, 40, 36)
    lab, cls = SY.planted_labels(shape, fill=0.2, rmin=4, rmax=9, seed=77, n_classes=3)
    thing, labels = [1, 2], [1, 2, 3]
    for name, ax in (('xy', 0), ('xz', 1), ('yz', 2)):
        heads = SY.planted_heads(lab, cls, name, n_classes=3, seed=13)
        pan, _ = panoptic_stack(heads['sem'].cuda(), heads['ctr_hmp'].cuda(), heads['offsets'].cuda(),
                                thing_list=thing, label_divisor=1000, stuff_area=16, void_label=0, nms_threshold=0.1,
                                nms_kernel=7, confidence_thr=0.5, median_kernel_size=3, coarse_boundaries=False)
        whole = PA.track_stack(pan, name, shape, labels, thing, 1000, 0.25, 0.25, as_tracks=True)
        assert whole.n_inst > 5
        D = pan.shape[0]
        for cut in (D // 2, 7):
            bounds = [0, cut, D]
            tabs, hosts = [], []
            for r in range(2):
                lo, hi = bounds[r], bounds[r + 1]
                ext = pan[lo:hi + 1] if r == 0 else pan[lo:hi]
                t, h = PA.tables_from_stack(ext.contiguous(), labels, thing, 1000)
                tabs.append(t); hosts.append(h)
            counts = np.array([cut, D - cut])
            merged, own = sharded.merge_rank_tables(hosts, counts)
            final, first_seen = PA.chain_from_tables(merged, D, labels, thing, 1000, 0.25, 0.25)
            parts = [DT.plane_tracks(tabs[r], merged, final, first_seen, name, shape, labels, 1000, slice0=bounds[r],
                                     local_comp_index=own[r]) for r in range(2)]
            for part in parts:
                np.testing.assert_array_equal(part.inst_label, whole.inst_label)
                np.testing.assert_array_equal(part.inst_box, whole.inst_box)
                np.testing.assert_array_equal(part.inst_area, whole.inst_area)
            key = torch.cat([part.key[:part.n_runs] for part in parts])
            ln = torch.cat([part.ln[:part.n_runs] for part in parts])
            k2, st2, l2, n2 = DT.sort_runs(key, ln, int(key.numel()), merge_touching=(name == 'yz'))
            assert n2 == whole.n_runs
            assert torch.equal(k2[:n2], whole.key[:n2]) and torch.equal(l2[:n2], whole.ln[:n2])
            assert torch.equal(st2[:n2], whole.st[:n2])
        # the materialised trackers against the whole-axis tracker objects
        solo = sharded.track_plane(pan, name, shape, labels, thing, 1000)
        for a, b in zip(solo.trackers(), whole.trackers()):
            assert_instances_equal(a.instances, b.instances)


def test_evaluator_on_tracker_jsons(tmp_path):
    """Evaluator over two tracker json files (evaluator.py:24-122): the RLE route (rle_matcher on run tables)
    and the voxel route (volume_pq: joint histogram of the filled volumes) must agree on PQ; F1/precision/recall
    follow from the same match."""
    from empanada_amd import evaluation as EV
    from empanada_amd.inference import patterns as PA
    shape = (20, 64, 72)
    lab_gt, _ = SY.planted_labels(shape, fill=0.2, rmin=4, rmax=10, seed=3)
    lab_pr = lab_gt.copy()
    ids = np.unique(lab_gt)[1:]
    lab_pr[lab_pr == ids[0]] = 0                                      # one miss
    lab_pr[:, :, 36:][lab_pr[:, :, 36:] == ids[1]] = 0                # one shrunk object
    lab_pr[0:2, 0:2, 0:2] = lab_gt.max() + 5                          # one false positive
    paths = []
    for name, lab in (('gt', lab_gt), ('pred', lab_pr)):
        pan = torch.from_numpy(np.where(lab > 0, 1000 + lab, 0).astype(np.int32)).cuda()
        tr = PA.track_stack(pan, 'xy', shape, [1], [1], 1000, 0.25, 0.25)[0]
        p = str(tmp_path / f'{name}.json')
        tr.write_to_json(p)
        paths.append(p)
    ev = EV.Evaluator(semantic_metrics={'iou': EV.iou},
                      instance_metrics={'f1_50': EV.f1_50, 'f1_75': EV.f1_75, 'precision_50': EV.precision_50,
                                        'recall_50': EV.recall_50, 'ap': EV.ap},
                      panoptic_metrics={'pq': EV.panoptic_quality})
    res, inst = ev(paths[0], paths[1], return_instances=True)
    pq_vox, n_gt, n_pr, n_match = EV.volume_pq(lab_gt, lab_pr)
    assert len(inst['gt_matched']) == n_match and len(inst['gt_unmatched']) == n_gt - n_match
    assert abs(res['pq'] - pq_vox) < 1e-12
    tp = int(np.count_nonzero(inst['matched_ious'] >= 0.5))
    fp = len(inst['pred_unmatched']); fn = len(inst['gt_unmatched'])
    assert res['f1_50'] == tp / (tp + 0.5 * fp + 0.5 * fn) and res['ap'] == tp / (tp + fp + fn)
    assert res['precision_50'] == tp / (tp + fp) and res['recall_50'] == tp / (tp + fn)
    assert res['f1_75'] <= res['f1_50'] < 1
    inter = np.count_nonzero((lab_gt > 0) & (lab_pr > 0)); union = np.count_nonzero((lab_gt > 0) | (lab_pr > 0))
    assert res['iou'] == inter / union
    assert ev(paths[0], paths[0])['pq'] == pytest.approx(1.0, abs=1e-4)


def test_data_post_fixture_of_the_reference_gpu():
    """The mask of the reference's tests/test_data_post.py (real organelle shapes, two stuff classes around seven
    things) through the HIP get_panoptic_segmentation: labels and centres identical to the reference's output."""
    from empanada_amd.inference.postprocess import get_panoptic_segmentation
    g = load_golden('data_post')
    sem = torch.from_numpy(g['sem'].astype(np.int64))[None, None].cuda()
    pan, ctr = get_panoptic_segmentation(sem, torch.from_numpy(g['ctr_hmp'])[None].cuda(),
                                         torch.from_numpy(g['offsets'])[None].cuda(), [2], 1000, 0, 0, 0.1, 7)
    np.testing.assert_array_equal(pan.cpu().numpy(), g['pan'])
    np.testing.assert_array_equal(ctr.cpu().numpy(), g['ctr'])


def test_get_panoptic_segmentation_golden():
    """P6 (postprocess.py:298-356): shape checks + centres + grouping + fusion in one call, against the reference"""
    from empanada_amd.inference.postprocess import get_panoptic_segmentation
    g = load_golden('panoptic_seg')
    for i in range(int(g['n'])):
        C, k, seed = (int(x) for x in g[f'c{i}_par'])
        thr = float(g[f'c{i}_thr'])
        lab, cls = SY.planted_labels((3, 72, 88), fill=0.25, rmin=4, rmax=10, seed=seed, n_classes=max(C - 1, 1))
        heads = SY.planted_heads(lab, cls, 'xy', n_classes=1 if C == 1 else C - 1, seed=seed)
        for z in range(3):
            prob = heads['sem'][z:z + 1]
            sem = (prob >= 0.5).long() if C == 1 else torch.argmax(prob, dim=1, keepdim=True)
            pan, ctr = get_panoptic_segmentation(sem.cuda(), heads['ctr_hmp'][z:z + 1].cuda(),
                                                 heads['offsets'][z:z + 1].cuda(), [1], 1000, 16, 0, thr, k)
            assert pan.dtype == torch.int64 and ctr.dtype == torch.int64
            np.testing.assert_array_equal(pan.cpu().numpy(), g[f'c{i}_z{z}_pan'], err_msg=f'{i} {z}')
            np.testing.assert_array_equal(ctr.cpu().numpy(), g[f'c{i}_z{z}_ctr'], err_msg=f'{i} {z}')
    with pytest.raises(ValueError):
        get_panoptic_segmentation(torch.zeros((1, 2, 8, 8)).cuda(), torch.zeros((1, 1, 8, 8)).cuda(),
                                  torch.zeros((1, 2, 8, 8)).cuda(), [1], 1000, 16, 0)
    with pytest.raises(ValueError):
        get_panoptic_segmentation(torch.zeros((2, 1, 8, 8), dtype=torch.long).cuda(), torch.zeros((2, 1, 8, 8)).cuda(),
                                  torch.zeros((2, 2, 8, 8)).cuda(), [1], 1000, 16, 0)


class _ListQueue:
    def __init__(self, items):
        self.items = list(items)

    def get(self):
        return self.items.pop(0)


class _Sink:
    sent = None

    def send(self, obj):
        self.sent = obj

    def close(self):
        pass


def test_forward_multigpu_golden():
    """patterns.forward_multigpu (patterns.py:279-350) fed through a queue like the matcher process of
    scripts/inference3d_multigpu.py: the rle_stack it sends equals the reference's"""
    from conftest import unpack_rle_seg
    from empanada_amd.inference import engines as EN
    from empanada_amd.inference import patterns as PA
    g = load_golden('forward_multigpu')
    for i in range(int(g['n'])):
        C, ks, seed, n_out = (int(x) for x in g[f'c{i}_par'])
        nthing = 1 if C == 1 else C - 1
        lab, cls = SY.planted_labels((9, 56, 64), fill=0.25, rmin=4, rmax=9, seed=seed, n_classes=nthing)
        heads = SY.planted_heads(lab, cls, 'xy', n_classes=nthing, seed=seed)
        labels = [1] if C == 1 else [1, 2]
        eng = EN.PanopticDeepLabRenderEngine(torch.nn.Identity(), thing_list=[1], label_divisor=1000, nms_kernel=7,
                                             nms_threshold=0.1, confidence_thr=0.5, coarse_boundaries=False)
        items = []
        for z in range(lab.shape[0]):
            cells = eng.get_instance_cells(heads['ctr_hmp'][z:z + 1].cuda(), heads['offsets'][z:z + 1].cuda())
            items.append((heads['sem'][z:z + 1].cuda(), cells))
        items.append(('finish', 'finish'))
        sink = _Sink()
        PA.forward_multigpu(PA.create_matchers([1], 1000, 0.25, 0.25), _ListQueue(items), [], sink, 0.5, ks, labels,
                            1000, [1], 16, 0)
        stack = sink.sent[0]
        assert len(stack) == n_out
        for z, rs in enumerate(stack):
            exp = unpack_rle_seg(g, f'c{i}_z{z}')
            for c in labels:
                assert_instances_equal(rs[c], exp.get(c, {}))


def test_logits_to_prob_gpu_vs_cpu_at_the_threshold():
    """D2 (engines.py:22-30) is torch on either side: sigmoid / softmax on the GPU against the same call on the CPU.
    What matters downstream is the hardening decision `p >= thr` (engines.py:114-121) and the argmax: checked on a
    dense sweep of logits around logit(thr) for the thresholds the configs use, and on random multi-class logits.
    The probabilities themselves may differ in the last place (different exp implementations); the bound is asserted."""
    from empanada_amd.inference.engines import logits_to_prob
    gen = torch.Generator().manual_seed(0)
    for thr in (0.3, 0.5):
        centre = float(np.log(thr / (1 - thr)))
        near = centre + (torch.rand((1, 1, 1024, 2048), generator=gen) - 0.5) * 2e-3
        exact = torch.full((1, 1, 1, 16), centre)
        wide = (torch.rand((1, 1, 1024, 2048), generator=gen) - 0.5) * 16
        for x in (near, exact, wide):
            cpu = logits_to_prob(x)
            gpu = logits_to_prob(x.cuda()).cpu()
            ulp = (gpu.view(torch.int32) - cpu.view(torch.int32)).abs().max().item()
            assert ulp <= 2, ulp
            flips = int(((gpu >= thr) != (cpu >= thr)).sum())
            # a flip needs |p - thr| below one ulp of p: record how often that happens on 2M logits within 1e-3
            assert flips <= 4, (thr, flips)
    x = torch.randn((1, 5, 512, 512), generator=gen) * 3
    cpu = logits_to_prob(x)
    gpu = logits_to_prob(x.cuda()).cpu()
    assert (gpu - cpu).abs().max().item() <= 5e-7
    assert int((gpu.argmax(dim=1) != cpu.argmax(dim=1)).sum()) == 0


def test_forward_is_deterministic_from_run_to_run():
    """with the hand-written kernels at every site that has one, the prepared model gives bit-identical heads on
    repeated calls: the library's split-K GEMM with atomic accumulation in the ASPP image-pooling branch (the source of
    run-to-run differences in bench.py's forward checksum) runs on the fused conv kernel.  (With MIOpen at every site
    the result depends on which of its kernels the find step picks: some accumulate with atomics.)"""
    from empanada_amd.models import PanopticDeepLab, prepare_for_inference, synthesize_weights
    from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
    m = prepare_for_inference(synthesize_weights(PanopticDeepLab(encoder='resnet50', num_classes=1)), 'cuda')
    x = torch.randn(4, 1, 256, 256, device='cuda').contiguous(memory_format=torch.channels_last)
    for force in ('direct',):
        if force:
            for mod in m.modules():
                if isinstance(mod, FusedConvBNAct) and force in mod.candidates(False):
                    mod.impl = force
        with torch.no_grad():
            a = {k: v.clone() for k, v in m(x).items()}
            for _ in range(2):
                b = m(x)
                for k in a:
                    assert torch.equal(a[k], b[k]), (force, k)


def _random_tracker_volumes(seed, shape=(36, 40, 44)):
    """three label volumes of the same scene as three imperfect observers would see it: objects dropped, shifted by a
    voxel or two, eroded, split in two, or fused with a neighbour -- so that pair IoUs fall on both sides of the
    cluster cut, hubs link several clusters and voted clusters overlap (every branch of the consensus)"""
    rng = np.random.default_rng(seed)
    base, _ = SY.planted_labels(shape, fill=0.3, rmin=3, rmax=8, seed=seed + 1000)
    n = int(base.max())
    vols = []
    for t in range(3):
        v = np.zeros(shape, dtype=np.int64)
        for i in range(1, n + 1):
            m = base == i
            if not m.any():
                continue
            mode = rng.integers(0, 8)
            if mode == 0:
                continue                                               # missed
            if mode == 1:
                m = np.roll(m, int(rng.integers(-2, 3)), axis=int(rng.integers(0, 3)))
            elif mode == 2:                                            # eroded along one axis
                ax = int(rng.integers(0, 3))
                m = m & np.roll(m, 1, axis=ax) & np.roll(m, -1, axis=ax)
            elif mode == 3:                                            # split in two by a gap
                idx = np.nonzero(m)
                ax = int(rng.integers(0, 3))
                cut = int(np.median(idx[ax]))
                sl = [slice(None)] * 3
                sl[ax] = cut
                m = m.copy()
                m[tuple(sl)] = False
            lab = 1000 + i
            if mode == 4 and i > 1:
                lab = 1000 + i - 1                                     # fused with the previous object's label
            v[m] = lab
        vols.append(v)
    return vols


@pytest.mark.parametrize('seed', range(12))
def test_consensus_random_scenes_equal_oracle(seed):
    """differential test of merge_objects_from_trackers (box screening, pair intersections, cluster graph incl. shared
    hubs, votes, overlap joins on the GPU + tables) against the oracle (pinned to the reference by its six KATs) on
    random imperfect observers, for the vote / IoU / bypass settings the reference's tests use"""
    from empanada_amd import consensus as CO
    from empanada_amd.inference import rle, tracker
    from oracle import consensus as OC
    from oracle import rle_seg as OS
    vols = _random_tracker_volumes(seed)
    shape = vols[0].shape
    trs, otrs = [], []
    for ti, v in enumerate(vols):
        tr = tracker.InstanceTracker(1, 1000, shape, axis='xy')
        otr = OS.InstanceTracker(1, 1000, shape, 'xy')
        segs, _ = rle.stack_to_rle_segs(torch.from_numpy(v.astype(np.int32)).cuda().view(torch.uint32), [1], 1000, [1],
                                        force_connected=True)
        for z in range(shape[0]):
            tr.update(segs[z][1], z)
            otr.update(OS.pan_seg_to_rle_seg(v[z], [1], 1000, [1], force_connected=True)[1], z)
        tr.finish()
        otr.finish()
        assert_instances_equal(tr.instances, otr.instances, check_order=False)
        trs.append(tr)
        otrs.append(otr)
    n_cases = 0
    for vote, iou_thr, bypass in ((2, 0.75, False), (2, 0.75, True), (1, 0.75, False), (3, 0.75, False), (2, 0.1, False),
                                  (1, 0.75, True), (2, 0.5, False)):
        try:
            exp = OC.merge_objects_from_trackers(otrs, vote, iou_thr, bypass)
        except Exception as e:                                     # the reference's single-range join failure etc.
            with pytest.raises(type(e)):
                CO.merge_objects_from_trackers(trs, vote, iou_thr, bypass)
            continue
        got = CO.merge_objects_from_trackers(trs, vote, iou_thr, bypass)
        assert_instances_equal(got, exp)
        n_cases += 1
    assert n_cases >= 4


def test_consensus_random_scenes_reach_every_branch():
    """(runs after the parametrised cases above) the random scenes exercised the array fast path, the literal
    cluster-graph path, hubs shared by several clusters and the overlap joins"""
    from empanada_amd import consensus as CO
    c = CO.BRANCH_COUNTS
    assert c['fast_components'] > 10 and c['general_components'] > 10, c
    assert c['extra_memberships'] > 0 and c['overlap_joins'] > 0, c


def _blobby_stack(seed, shape=(20, 36, 40), n=45, div=1000):
    """random boxes that move, split and merge from slice to slice and touch each other (also across row ends):
    instances made of several components per slice, runs of one instance that are contiguous in flat index, labels
    that merge through the IoA rule -- what the planted ellipsoids rarely produce"""
    rng = np.random.default_rng(seed)
    D, H, W = shape
    pan = np.zeros(shape, dtype=np.int64)
    for _ in range(n):
        z0, z1 = sorted(rng.integers(0, D, 2))
        y, x = rng.integers(0, H - 8), rng.integers(0, W - 8)
        h, w = rng.integers(3, 12), rng.integers(3, 12)
        for z in range(z0, z1 + 1):
            y = int(np.clip(y + rng.integers(-2, 3), 0, H - h))
            x = int(np.clip(x + rng.integers(-2, 3), 0, W - w))
            pan[z, y:y + h, x:x + w] = div + 1 + rng.integers(0, 3)
            if rng.random() < 0.2:
                pan[z, y + h // 2, x:x + w] = 0                     # split
            if rng.random() < 0.1:
                pan[z, y, :] = div + 1 + rng.integers(0, 3)           # a full-width row: runs wrap over the row end
    return pan


@pytest.mark.parametrize('seed', range(6))
@pytest.mark.parametrize('axis', ['xy', 'xz', 'yz'])
def test_track_stack_random_stacks_equal_oracle(seed, axis):
    """differential test of the whole per-plane chain on the device (runs, components, reduced overlaps, native chain
    with the native Hungarian step, run lift + merge + sort, materialised trackers) against the oracle's per-slice
    protocol (pan_seg_to_rle_seg -> forward / backward matching -> InstanceTracker.update / finish), all three planes"""
    from empanada_amd.inference import patterns as PA
    from oracle import rle_seg as OS
    pan = _blobby_stack(seed)
    D, H, W = pan.shape
    shape3d = {'xy': (D, H, W), 'xz': (H, D, W), 'yz': (H, W, D)}[axis]
    trs = PA.track_stack(torch.from_numpy(pan.astype(np.int32)).cuda().view(torch.uint32), axis, shape3d, [1], [1], 1000,
                         0.25, 0.25)
    matchers = OS.create_matchers([1], 1000, 0.25, 0.25)
    stack = OS.forward_matching([pan[z] for z in range(D)], matchers, [1], 1000, [1])
    otr = OS.create_axis_trackers([axis], [1], 1000, shape3d)[axis]
    for idx, rs in OS.backward_matching(stack, matchers, D):
        OS.update_trackers(rs, idx, otr)
    OS.finish_tracking(otr)
    assert len(otr[0].instances) > 5
    assert_instances_equal(trs[0].instances, otr[0].instances)
