#!/usr/bin/env python
"""bench.py -- end-to-end 3D panoptic inference throughput on MI355X (Mvox/s).

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).
A "step" is one full pass of the hot path over one synthetic volume: model forward over every slice
(PyTorch-ROCm, fp32, the named architecture with synthesised weights) -> sigmoid -> recursive median +
harden -> centres -> pixel grouping -> semantic/instance fusion -> runs + 8-connected components ->
slice-to-slice overlaps -> forward/backward label propagation -> trackers -> size/span filters -> labelled
uint32 volume written to (pinned) host memory.

Workload at N=1 = BASELINE.json configs[1]: single-GPU stack inference, 256x512x512 volume, ResNet-50 encoder.
Inputs are resident in HBM when the timed region starts (uint8 EM volume + planted head tensors, see
empanada_amd/synthetic.py and DESIGN.md "Synthetic workload": the conv forward is computed and timed on every
slice, its outputs are checksummed, and the post-processing consumes the planted heads so that it sees a
realistic object load; random weights would give it an empty or degenerate segmentation).

N>1 (launched by torch.distributed.run, one rank per GPU, RCCL): the volume's depth grows with N (weak
scaling, 256 slices per rank); see DESIGN.md "Multi-GPU".
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENGINE = dict(thing_list=[1], label_divisor=20000, stuff_area=64, void_label=0, nms_threshold=0.1, nms_kernel=7,
              confidence_thr=0.3, median_kernel_size=7)      # projects/mitonet/configs/mmm_median_inference.yaml
LABELS = [1]                                                  # --things T: classes 1..T, all of them things
MATCH = dict(merge_iou_thr=0.25, merge_ioa_thr=0.25)
FILTERS = dict(min_size=500, min_span=4)
NORM = dict(mean=0.508979, std=0.148561)                      # MitoNet norms
# Algorithmic HBM bytes per voxel of the single-kernel ABI calls, C = 1, full-res heads.
# f = fraction of voxels whose class is a thing (measured on the run's own data): only those read offsets.
ALG_BYTES = {
    'emp_median_harden_stack': lambda f: 4 * (1 if len(LABELS) == 1 else len(LABELS) + 1) + 1,  # prob fp32 x C, sem u8
    'emp_find_centers': lambda f: 4,                   # read heatmap
    'emp_group_pixels': lambda f: 1 + 8 * f + 2,       # read sem u8, offsets of voted pixels, write ids u16
    'emp_fuse_apply': lambda f: 1 + 2 + 4,             # read sem u8 + ids u16, write pan u32
    'emp_runs_count': lambda f: 4,                     # read pan
    'emp_runs_extract': lambda f: 4,                   # read pan (+12 B per run)
}
# HBM traffic per voxel from rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, calibrated on known byte counts;
# profiles/r1_pmc_postproc_256x512x512.md), same workload.  Collected offline: --pmc cannot run inside this script.
PMC_TRAFFIC_BYTES_PER_VOXEL = {
    'emp_median_harden_stack': 5.02, 'emp_find_centers': 4.56, 'emp_group_pixels': 4.60, 'emp_fuse_apply': 7.01,
    'emp_runs_count': 4.02, 'emp_runs_extract': 4.15,
}
# same, as a ratio to the algorithmic bytes, for the dense-path kernels whose shapes vary from call to call
# (profiles/r1_pmc_dense.md)
PMC_TRAFFIC_RATIO = {
    # HBM traffic (2 x FETCH_SIZE + WRITE_SIZE) / algorithmic bytes, averaged over the launches of a timed pass with
    # the tuned implementation choices replayed (--load-tune): profiles/r1_pmc_bench_dense.md
    'emp_conv_bn_act_nhwc': 1.18, 'emp_bn_act_nhwc': 1.0, 'emp_dwconv_nhwc': 1.06, 'emp_upsample_bilinear': 1.2,
}
DENSE_KERNELS = ('emp_bn_act_nhwc', 'emp_dwconv_nhwc', 'emp_upsample_bilinear', 'emp_conv_bn_act_nhwc',
                 'emp_wino_input_transform', 'emp_gemm_nt_batched', 'emp_wino_gemm_fused',
                 'emp_wino_output_transform', 'emp_wino4_input_transform', 'emp_wino4_output_transform', 'emp_wino3_input_transform',
                 'emp_wino3_output_transform',
                 'emp_pointwise_out_nhwc', 'emp_bn_relu_maxpool_nhwc', 'emp_slices_to_input')
MFMA_KERNELS = ('emp_conv_bn_act_nhwc', 'emp_gemm_nt_batched', 'emp_wino_gemm_fused')
MFMA_F32_PEAK_TFLOPS = 157.3                   # dense fp32 matrix peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0                          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--depth', type=int, default=256, help='slices per rank')
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--batch', type=int, default=128,
                    help='slices of 512 x 512 per model call; larger slices get proportionally fewer per call '
                         '(same pixels per call: the largest activations stay at 2 GiB)')
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16', 'fp16'])
    ap.add_argument('--model', default='pdl_r50', choices=sorted(MODELS),
                    help='pdl_r50 is the configuration the metric is quoted on; the others are side measurements')
    ap.add_argument('--tune-batch', type=int, default=32, help='slices of 512 x 512 the conv tuner times each site on')
    ap.add_argument('--things', type=int, default=1,
                    help='thing classes (1 = binary MitoNet, the headline; T > 1 = softmax over background + T classes)')
    ap.add_argument('--save-tune', default=None, help='write the tuned conv implementation per call site (json)')
    ap.add_argument('--load-tune', default=None, help='replay conv implementations from a --save-tune file')
    ap.add_argument('--conv-impls', default=None,
                    help='comma list restricting the tuner, e.g. miopen,direct (no Winograd forms)')
    ap.add_argument('--no-tune', action='store_true', help='keep MIOpen + epilogue pass for every convolution')
    ap.add_argument('--cpu-slices', type=int, default=96, help='slices of the same workload for the CPU baseline')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-pipeline', action='store_true', help='run the passes strictly one after the other')
    ap.add_argument('--mode', default='stack', choices=['stack', 'orthoplane'],
                    help='stack = BASELINE configs[1] (xy only); orthoplane = configs[2] (xy/xz/yz + consensus, '
                         'cubic volume of side --size, single GPU)')
    return ap.parse_args()


def build_inputs(D, S, device, seed_offset=0, things=1):
    from empanada_amd import synthetic as SY
    shape = (D, S, S)
    from empanada_amd.data import DeviceVolume
    vol = DeviceVolume(SY.em_volume(shape, seed=1234 + seed_offset), NORM['mean'], NORM['std'], 16, device)
    lab, cls = SY.planted_labels(shape, fill=0.08, rmin=6, rmax=24, seed=4321 + seed_offset, n_classes=things)
    heads = {'sem': [], 'ctr_hmp': [], 'offsets': []}
    for s in range(0, D, 64):                    # chunked to bound the generator's temporaries
        h = SY.planted_heads(lab, cls, 'xy', device=device, slices=slice(s, min(D, s + 64)), seed=99 + s,
                             n_classes=things)
        for k in heads:
            heads[k].append(h[k])
    heads = {k: torch.cat(v, dim=0).contiguous() for k, v in heads.items()}
    return vol, heads, int(cls.shape[0] - 1)


# --model name -> label in config.workload (the FLOP count below is only known for the headline model)
MODELS = {'pdl_r50': 'PanopticDeepLab/ResNet-50', 'bifpn_r50': 'PanopticBiFPN/ResNet-50',
          'bifpn_regnety': 'PanopticBiFPN/RegNetY-6.4GF'}


def build_model(name):
    from empanada_amd.models import PanopticBiFPN, PanopticDeepLab, synthesize_weights
    nc = 1 if len(LABELS) == 1 else len(LABELS) + 1              # binary head, or background + T classes
    if name == 'pdl_r50':
        model = PanopticDeepLab(encoder='resnet50', num_classes=nc)
    else:
        model = PanopticBiFPN(encoder={'bifpn_r50': 'resnet50', 'bifpn_regnety': 'regnety_6p4gf'}[name], num_classes=nc)
    return synthesize_weights(model)


class Pipeline:
    def __init__(self, args, device):
        from empanada_amd.models import prepare_for_inference
        self.dtype = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}[args.dtype]
        model = build_model(args.model)
        with torch.no_grad():                     # O(1) logits like a trained model (synthetic He weights are hot)
            for head in (model.semantic_head, model.ins_center, model.ins_xy):
                head.head[1].weight.mul_(1e-3)
        self.model = prepare_for_inference(model, device, self.dtype)
        self.device = device
        self.batch = args.batch
        self.tune_batch = args.tune_batch
        self.timers = {}
        self.tuned = {}
        self.conv_impls = args.conv_impls.split(',') if args.conv_impls else None
        self.post_stream = torch.cuda.Stream(device=device)

    def slices_per_call(self, h, w):
        return max(1, self.batch * 512 * 512 // max(h * w, 1))

    @torch.no_grad()
    def tune(self, size, save=None, load=None):
        """warm-up only: let every conv + BN call site pick its fastest implementation on the bench shapes
        (or replay the choices of an earlier run: profiler passes distort the timings the tuner relies on)"""
        from empanada_amd.models import tune_fused_convs
        from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
        if self.dtype != torch.float32:
            return
        counts = {}
        if load:
            choice = json.load(open(load))
            for name, m in self.model.named_modules():
                if isinstance(m, FusedConvBNAct):
                    m.impl = choice.get(name, 'miopen')
                    counts[m.impl] = counts.get(m.impl, 0) + 1
            self.tuned = counts
            log(f'conv call sites loaded from {load}: {counts}')
            return
        # tuned on a quarter of a call's slices: the ranking of the hand-written forms does not change with the batch,
        # and MIOpen's exhaustive find on the full-size shapes of all 66 sites would take minutes of warm-up
        n = max(1, min(self.slices_per_call(size, size), self.tune_batch * 512 * 512 // (size * size)))
        x = torch.rand((n, 1, size, size), device=self.device).contiguous(memory_format=torch.channels_last)
        rep = tune_fused_convs(self.model, x, allow=self.conv_impls)
        for _, (best, _) in rep.items():
            counts[best] = counts.get(best, 0) + 1
        self.tuned = counts
        saved = sum(t['miopen'] - min(t.values()) for _, t in rep.values())
        log(f'conv call sites tuned: {counts}; isolated saving {saved:.2f} ms per {n} slices')
        if save:
            json.dump({k: v[0] for k, v in rep.items()}, open(save, 'w'), indent=1)

    @torch.no_grad()
    def forward(self, dv, axis='xy', lo=0, hi=None):
        """slices [lo, hi) of one plane of the resident uint8 volume (empanada_amd.data.DeviceVolume: strided
        gather + normalise + pad in one HIP pass) -> resident sem probabilities (n,1,h,w) fp32 + a checksum of all
        heads"""
        hi = dv.n_slices(axis) if hi is None else hi
        h, w = dv.plane_shape(axis)
        nc = 1 if len(LABELS) == 1 else len(LABELS) + 1
        prob = torch.empty((hi - lo, nc, h, w), dtype=torch.float32, device=self.device)
        chk = torch.zeros((), dtype=torch.float64, device=self.device)
        for s, x in dv.batches(axis, self.slices_per_call(h, w), lo, hi):
            if self.dtype != torch.float32:
                x = x.to(self.dtype)
            out = self.model(x.contiguous(memory_format=torch.channels_last))
            logits = out['sem_logits'][..., :h, :w].float()          # logits_to_prob, engines.py:22-30
            prob[s - lo:s - lo + x.shape[0]] = torch.sigmoid(logits) if nc == 1 else torch.softmax(logits, dim=1)
            chk += out['ctr_hmp'].float().sum(dtype=torch.float64) + out['offsets'].float().sum(dtype=torch.float64)
        return prob, chk + prob.sum(dtype=torch.float64)

    def postprocess(self, heads, out_host):
        """probabilities -> labelled slab in pinned host memory.  Same code path for 1 and N ranks
        (empanada_amd/inference/sharded.py); with one rank the collectives are no-ops."""
        from empanada_amd.inference import sharded
        t0 = time.perf_counter()
        pan = sharded.sharded_panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'],
                                             coarse_boundaries=False, **ENGINE)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        vol = sharded.sharded_stack_volume(pan, LABELS, ENGINE['thing_list'], ENGINE['label_divisor'],
                                           min_size=FILTERS['min_size'], min_span=FILTERS['min_span'], **MATCH)
        out_host.copy_(vol.view(torch.int32), non_blocking=True)
        t2 = time.perf_counter()
        self.timers.setdefault('stages', []).append(
            {'panoptic_stack_incl_wait_for_forward': t1 - t0, 'runs_chain_fill_launch': t2 - t1})
        return vol


def build_inputs_ortho(S, device, rank=0, world=1):
    """cubic volume; every rank holds the EM slices and planted heads of its own contiguous block per plane"""
    from empanada_amd import synthetic as SY
    from empanada_amd.inference.sharded import shard_bounds
    shape = (S, S, S)
    b = shard_bounds(S, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    from empanada_amd.data import DeviceVolume
    dv = DeviceVolume(SY.em_volume(shape, seed=1234), NORM['mean'], NORM['std'], 16, device)
    lab, cls = SY.planted_labels(shape, fill=0.08, rmin=6, rmax=24, seed=4321)
    heads, stacks = {}, {}
    for axis, ax in (('xy', 0), ('xz', 1), ('yz', 2)):
        stacks[axis] = (dv, axis, lo, hi)          # the rank's block of the plane: a strided view of the volume
        parts = {'sem': [], 'ctr_hmp': [], 'offsets': []}
        for s in range(lo, hi, 64):
            h = SY.planted_heads(lab, cls, axis, device=device, slices=slice(s, min(hi, s + 64)), seed=99 + s)
            for k in parts:
                parts[k].append(h[k])
        heads[axis] = {k: torch.cat(v, dim=0).contiguous() for k, v in parts.items()}
    return stacks, heads, int(cls.shape[0] - 1), lo


def orthoplane_step(pipe, stacks, heads, slice0, shape3d, host_out, stages, first=None, prefetch_next=False):
    """One pass of BASELINE configs[2]: three slice-sharded stacks (xy, xz, yz) -> trackers stitched on rank 0 ->
    filters -> instance consensus -> filters -> labelled volume in pinned host memory
    (scripts/pdl_inference3d.py:110-233 in orthoplane mode).
    Consecutive passes are software-pipelined like the stack mode: with prefetch_next the xy forward of the NEXT pass
    is queued as soon as this pass's yz tables are on the host, so the tail (yz tracking, consensus, fill, D2H; post
    stream + host) runs under it; the caller hands the returned (checksum, event) back as `first`."""
    from empanada_amd.inference import sharded
    trackers = {}
    chk = 0
    # Two HIP streams.  The forward of plane p+1 is queued (default stream) as soon as the device tables of plane p
    # are on the host; the host half of plane p (label-propagation chain, tracker assembly) and its device work
    # (D2H copies, yz scatter) run on the post-processing stream meanwhile, so neither side waits for the other.
    # (Queuing all three forwards up front does not work: ~14k launches exceed the HIP queue and the host blocks.)
    post = pipe.post_stream
    planes = ('xy', 'xz', 'yz')
    if first is None:
        prob, c = pipe.forward(*stacks['xy'])
        ev = torch.cuda.Event()
        ev.record()
    else:
        c, ev = first
    chk = chk + c
    nxt = None
    with torch.cuda.stream(post):
        for i, axis in enumerate(planes):
            t0 = time.perf_counter()
            post.wait_event(ev)
            h = heads[axis]
            pan = sharded.sharded_panoptic_stack(h['sem'], h['ctr_hmp'], h['offsets'], coarse_boundaries=False,
                                                 **ENGINE)
            table, host = sharded.sharded_tables(pan, [1], ENGINE['thing_list'], ENGINE['label_divisor'])
            t1 = time.perf_counter()
            if i + 1 < len(planes):
                with torch.cuda.stream(torch.cuda.default_stream()):
                    prob, c = pipe.forward(*stacks[planes[i + 1]])
                    chk = chk + c
                    ev = torch.cuda.Event()
                    ev.record()
            elif prefetch_next:
                with torch.cuda.stream(torch.cuda.default_stream()):
                    prob, c2 = pipe.forward(*stacks['xy'])
                    ev2 = torch.cuda.Event()
                    ev2.record()
                    nxt = (c2, ev2)
            t2 = time.perf_counter()
            trackers[axis] = sharded.finish_plane(table, host, pan.shape[0], axis, shape3d, slice0, [1],
                                                  ENGINE['thing_list'], ENGINE['label_divisor'], **MATCH)
            stages[f'{axis}_wait_forward_pixels_tables'] = stages.get(f'{axis}_wait_forward_pixels_tables', 0) + t1 - t0
            stages[f'{axis}_enqueue_next_forward'] = stages.get(f'{axis}_enqueue_next_forward', 0) + t2 - t1
            stages[f'{axis}_tracking'] = stages.get(f'{axis}_tracking', 0) + time.perf_counter() - t2
        n_found = _orthoplane_finish(trackers, shape3d, host_out, stages)
    torch.cuda.current_stream().wait_stream(post)
    return chk, n_found, nxt


def _orthoplane_finish(trackers, shape3d, host_out, stages):
    from empanada_amd.inference import sharded
    n_found = 0
    if trackers['xy'] is not None:                   # rank 0 holds the stitched trackers
        t0 = time.perf_counter()
        cons, vols = sharded.consensus_volume(trackers, shape3d, [1], ENGINE['thing_list'], 2, 0.75, False,
                                              FILTERS['min_size'], FILTERS['min_span'])
        t1 = time.perf_counter()
        host_out.copy_(vols[1].view(torch.int32), non_blocking=True)
        torch.cuda.current_stream().synchronize()          # the post stream only: a prefetched forward keeps running
        stages['consensus_and_fill'] = stages.get('consensus_and_fill', 0) + t1 - t0
        stages['to_host'] = stages.get('to_host', 0) + time.perf_counter() - t1
        n_found = len(cons[1].instances)
    return n_found


def main_orthoplane(args, device, rank, world):
    import torch.distributed as dist
    S = args.size
    log(f'orthoplane: building inputs {S}^3 (rank {rank}/{world})')
    stacks, heads, n_obj, slice0 = build_inputs_ortho(S, device, rank, world)
    log(f'inputs ready ({n_obj} planted objects)')
    pipe = Pipeline(args, device)
    if not args.no_tune:
        pipe.tune(S, args.save_tune, args.load_tune)
    shape3d = (S, S, S)
    host_out = torch.empty(shape3d, dtype=torch.int32).pin_memory() if rank == 0 else None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        orthoplane_step(pipe, stacks, heads, slice0, shape3d, host_out, {})
        log(f'warmup {i} done')
    barrier()
    stages = {}
    t0 = time.perf_counter()
    first = None
    for k in range(args.steps):
        chk, n_found, first = orthoplane_step(pipe, stacks, heads, slice0, shape3d, host_out, stages, first,
                                              prefetch_next=(k + 1 < args.steps) and not args.no_pipeline)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    res = {
        'metric': 'Mvox/s end-to-end 3D panoptic inference (incl. consensus); PQ vs CPU ref',
        'value': round(float(S) ** 3 * args.steps / dt / 1e6, 3), 'unit': 'Mvox/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 2),
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f32' if args.dtype == 'fp32' else args.dtype, 'data': 'synthetic',
        'config': {'workload': f'orthoplane (xy/xz/yz) inference + instance consensus, {S}^3 uint8 volume, '
                               f'{MODELS[args.model]} C=1 forward on every slice of every plane + HIP '
                               f'post-processing on planted heads, {n_obj} planted objects, slices sharded over '
                               f'{world} rank(s)',
                   'mode': 'orthoplane', 'objects_found': int(n_found)},
        'stages_s_per_step': {k: round(v / args.steps, 4) for k, v in stages.items()},
    }
    print(json.dumps(res), flush=True)


def cpu_baseline(args, vol_u8, heads, n_slices):
    """The oracle chain (CPU restatement of the reference) + torch-CPU forward on a bounded sample of the
    same workload: the first n_slices slices.  kind = 'port'."""
    from oracle import postprocess as OP
    from oracle import rle_ops as OR
    from oracle import rle_seg as OS
    n = min(n_slices, vol_u8.shape[0])
    cores = min(16, os.cpu_count() or 1)          # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    model = build_model(args.model).eval()
    x = vol_u8[:n].cpu().float().unsqueeze(1)
    x = (x - 255 * NORM['mean']) / (255 * NORM['std'])
    sem, ctr, off = (heads[k][:n].cpu().numpy() for k in ('sem', 'ctr_hmp', 'offsets'))
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(n):
            out = model(x[i:i + 1])
            _ = torch.sigmoid(out['sem_logits']) if len(LABELS) == 1 else torch.softmax(out['sem_logits'], dim=1)
    t_conv = time.perf_counter() - t0
    pans = OP.engine3d_stack([sem[t:t + 1] for t in range(n)], [ctr[t:t + 1] for t in range(n)],
                             [off[t:t + 1] for t in range(n)], coarse_boundaries=False, render=True, **ENGINE)
    pans = [p.squeeze() for p in pans]
    matchers = OS.create_matchers(ENGINE['thing_list'], ENGINE['label_divisor'], MATCH['merge_iou_thr'],
                                  MATCH['merge_ioa_thr'])
    stack = OS.forward_matching(pans, matchers, LABELS, ENGINE['label_divisor'], ENGINE['thing_list'])
    shape = (len(pans),) + pans[0].shape
    trs = OS.create_axis_trackers(['xy'], LABELS, ENGINE['label_divisor'], shape)['xy']
    for idx, rs in OS.backward_matching(stack, matchers, len(pans)):
        OS.update_trackers(rs, idx, trs)
    OS.finish_tracking(trs)
    for tr in trs:
        OS.remove_small_objects(tr, FILTERS['min_size'])
        OS.remove_pancakes(tr, FILTERS['min_span'])
    out = np.zeros(shape, dtype=np.uint32)
    for tr in trs:
        OR.numpy_fill_instances(out, tr.instances)
    dt = time.perf_counter() - t0
    vox = float(np.prod(shape))
    # the HIP path on exactly the same sample -> PQ against the CPU result and identity of the instance ids
    from empanada_amd.evaluation import volume_pq
    from empanada_amd.inference import sharded
    sub = {k: heads[k][:n].contiguous() for k in heads}
    pan = sharded.sharded_panoptic_stack(sub['sem'], sub['ctr_hmp'], sub['offsets'], coarse_boundaries=False, **ENGINE)
    got = sharded.sharded_stack_volume(pan, LABELS, ENGINE['thing_list'], ENGINE['label_divisor'],
                                       min_size=FILTERS['min_size'], min_span=FILTERS['min_span'], **MATCH)
    got = got.view(torch.int32).cpu().numpy().astype(np.uint32)
    pq, n_gt, n_pred, n_match = volume_pq(out, got)
    return {'value': round(vox / dt / 1e6, 4), 'unit': 'Mvox/s', 'cores': cores, 'kind': 'port',
            'sample': f'first {len(pans)} of {vol_u8.shape[0]} slices ({shape[1]}x{shape[2]}), same heads; '
                      f'conv {t_conv:.1f}s of {dt:.1f}s', 'objects': int(sum(len(t.instances) for t in trs)),
            'pq_vs_cpu_ref': round(pq, 6), 'ids_identical': bool(np.array_equal(out, got)),
            'instances_cpu_gpu_matched': [n_gt, n_pred, n_match]}


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    import torch.distributed as dist
    backend = os.environ.get('EMP_BENCH_BACKEND', 'nccl')     # 'gloo': rehearse N ranks on fewer GPUs (not a measurement)
    if backend == 'gloo':
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group(backend)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    from empanada_amd import _hip
    _hip.load()
    torch.backends.cudnn.benchmark = True

    if args.things > 1:
        assert args.mode == 'stack', '--things > 1 is implemented for the stack mode'
        LABELS[:] = list(range(1, args.things + 1))
        ENGINE['thing_list'] = list(LABELS)
    if args.mode == 'orthoplane':
        main_orthoplane(args, device, rank, world)
        if world > 1:
            dist.destroy_process_group()
        return
    D, S = args.depth, args.size
    log(f'building inputs {D}x{S}x{S}')
    vol, heads, n_obj = build_inputs(D, S, device, seed_offset=rank, things=args.things)
    log(f'inputs ready ({n_obj} planted objects); building model')
    pipe = Pipeline(args, device)
    if not args.no_tune:
        pipe.tune(S, args.save_tune, args.load_tune)
    host_out = torch.empty((D, S, S), dtype=torch.int32).pin_memory()
    shape3d = (D, S, S)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        t_w = time.perf_counter()
        prob, chk = pipe.forward(vol)
        torch.cuda.synchronize()
        log(f'warmup {i}: forward {time.perf_counter() - t_w:.2f}s')
        out = pipe.postprocess(heads, host_out)
        torch.cuda.synchronize()
        log(f'warmup {i}: total {time.perf_counter() - t_w:.2f}s')
    barrier()
    _hip.PROFILE = {}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * args.steps)]
    from empanada_amd.inference import sharded
    t0 = time.perf_counter()
    if args.no_pipeline:
        for k in range(args.steps):
            ev[3 * k].record()
            prob, chk = pipe.forward(vol)
            ev[3 * k + 1].record()
            out = pipe.postprocess(heads, host_out)
            ev[3 * k + 2].record()
    else:
        # Software pipeline over consecutive passes on two HIP streams.  The forward of pass k is queued on the
        # default stream first; everything downstream of pass k-1's forward (pixel kernels, run tables and their
        # D2H, the host label-propagation chain, fill, D2H of the labelled slab) then runs on the post-processing
        # stream behind an event of forward k-1, concurrently with forward k.  The host never blocks the forward
        # queue, and all K passes complete inside the timed region (drain iteration + barrier below).
        post = pipe.post_stream
        fwd_done = [torch.cuda.Event() for _ in range(args.steps)]

        def downstream(k):
            with torch.cuda.stream(post):
                post.wait_event(fwd_done[k])
                pan = sharded.sharded_panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'],
                                                     coarse_boundaries=False, **ENGINE)
                table, host = sharded.sharded_tables(pan, LABELS, ENGINE['thing_list'], ENGINE['label_divisor'])
                tc = time.perf_counter()
                final = sharded.gather_tables_and_chain(host, pan.shape[0], LABELS, ENGINE['thing_list'],
                                                        ENGINE['label_divisor'], min_size=FILTERS['min_size'],
                                                        min_span=FILTERS['min_span'], **MATCH)
                pipe.timers.setdefault('chain_s', []).append(time.perf_counter() - tc)
                out = sharded.fill_slab(table, final, tuple(pan.shape))
                host_out.copy_(out.view(torch.int32), non_blocking=True)
                ev[3 * k + 2].record()
            return out

        for k in range(args.steps + 1):
            if k < args.steps:
                ev[3 * k].record()
                prob, chk = pipe.forward(vol)                      # asynchronous: only enqueues
                ev[3 * k + 1].record()
                fwd_done[k].record()
                _hip.PROFILE_SKIP.update(DENSE_KERNELS)            # the dense-path calls are sampled in pass 0
            if k > 0:
                out = downstream(k - 1)
        torch.cuda.current_stream().wait_stream(post)
    barrier()
    dt = time.perf_counter() - t0
    log(f'timed {args.steps} steps in {dt:.2f}s')
    prof, _hip.PROFILE = _hip.PROFILE, None
    _hip.PROFILE_SKIP.clear()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        vox_total = float(D) * S * S * world * args.steps
        ms_step = dt / args.steps * 1e3
        fwd_ms = np.mean([ev[3 * k].elapsed_time(ev[3 * k + 1]) for k in range(args.steps)])
        post_ms = np.mean([ev[3 * k + 1].elapsed_time(ev[3 * k + 2]) for k in range(args.steps)])
        if len(LABELS) == 1:
            thing_frac = float((heads['sem'] >= ENGINE['confidence_thr']).float().mean().item())
        else:
            thing_frac = float((heads['sem'].argmax(dim=1) > 0).float().mean().item())
        vox = float(D) * S * S
        # per ABI call: launches, total ms and algorithmic bytes inside the timed region.  The per-voxel kernels
        # process the whole slab in one launch (ALG_BYTES x voxels); the dense-path kernels report their bytes
        # per call (shapes vary by layer) and are sampled during the first timed pass.
        stat = {}
        for name, evs in prof.items():
            ms = [e[0].elapsed_time(e[1]) for e in evs]
            by = [e[2] if e[2] is not None else (ALG_BYTES[name](thing_frac) * vox if name in ALG_BYTES else None)
                  for e in evs]
            fl = [e[3] for e in evs]
            passes = 1 if (name in DENSE_KERNELS and not args.no_pipeline) else args.steps
            stat[name] = {'calls_per_pass': len(ms) / passes, 'ms_per_pass': float(np.sum(ms)) / passes,
                          'avg_ms': float(np.mean(ms)),
                          'bytes': float(np.sum(by)) if all(b is not None for b in by) else None,
                          'flops': float(np.sum(fl)) if all(f is not None for f in fl) else None}
        kern = {k: v['avg_ms'] for k, v in stat.items()}
        per_call = {k: round(v, 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1])}
        per_pass = {k: round(v['ms_per_pass'], 3) for k, v in sorted(stat.items(), key=lambda kv: -kv[1]['ms_per_pass'])}
        roofs = {k: round(v['bytes'] / (v['avg_ms'] * len(prof[k]) * 1e-3) / 1e9, 1)
                 for k, v in stat.items() if v['bytes'] is not None}
        tflops = {k: round(v['flops'] / (v['avg_ms'] * len(prof[k]) * 1e-3) / 1e12, 1)
                  for k, v in stat.items() if v['flops'] is not None}
        # dominant hand-written kernel = most GPU time per pass among the calls with a byte / flop model
        dom = max((k for k in stat if stat[k]['bytes'] is not None), key=lambda k: stat[k]['ms_per_pass'])
        n_dom = len(prof[dom])
        alg = stat[dom]['bytes'] / n_dom                                  # algorithmic bytes per launch (mean)
        if dom in PMC_TRAFFIC_BYTES_PER_VOXEL:
            traffic = round(PMC_TRAFFIC_BYTES_PER_VOXEL[dom] * vox)
        elif PMC_TRAFFIC_RATIO.get(dom) is not None:
            traffic = round(PMC_TRAFFIC_RATIO[dom] * alg)
        else:
            traffic = None
        if dom in MFMA_KERNELS:
            roof = {'bound': 'mfma', 'kernel': dom, 'achieved': tflops[dom], 'peak': MFMA_F32_PEAK_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(tflops[dom] / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': traffic,
                    'alg_flops_per_launch': round(stat[dom]['flops'] / n_dom)}
        else:
            roof = {'bound': 'hbm', 'kernel': dom, 'achieved': roofs[dom], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(roofs[dom] / HBM_PEAK_GBS, 4), 'traffic': traffic}
        roof.update({'alg_bytes_per_launch': round(alg), 'avg_launch_ms': round(kern[dom], 4),
                     'launches_per_pass': stat[dom]['calls_per_pass'],
                     'ms_per_pass': round(stat[dom]['ms_per_pass'], 3), 'thing_fraction': round(thing_frac, 4),
                     'alg_bytes_per_voxel': {k: round(f(thing_frac), 3) for k, f in ALG_BYTES.items()},
                     'all_kernels_GBps': roofs, 'mfma_kernels_TFLOPs': tflops})
        flops = 414477.0 * D * S * S                                      # PDL-R50, C=1
        res = {
            'metric': 'Mvox/s end-to-end 3D panoptic inference (incl. consensus); PQ vs CPU ref',
            'value': round(vox_total / dt / 1e6, 3), 'unit': 'Mvox/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_step, 2), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32' if args.dtype == 'fp32' else args.dtype, 'data': 'synthetic',
            'config': {'workload': f'stack (xy) inference, {D * world}x{S}x{S} uint8 volume, {MODELS[args.model]} '
                                   f'C={1 if len(LABELS) == 1 else len(LABELS) + 1} fp-forward on every slice + HIP post-processing on planted heads '
                                   f'(ks=7, full-res heads), {n_obj} planted objects per rank',
                       'mode': 'stack', 'slices_per_rank': D, 'batch': pipe.slices_per_call(S, S),
                       'objects_found': int(len(np.unique(host_out.numpy())) - 1)},
            'breakdown_ms': {'forward': round(float(fwd_ms), 2), 'forward_end_to_slab_on_host': round(float(post_ms), 2),
                             'forward_TFLOPs': round(flops / (fwd_ms * 1e-3) / 1e12, 2) if args.model == 'pdl_r50' and len(LABELS) == 1 else None,
                             'pipelined': not args.no_pipeline, 'conv_impls': pipe.tuned,
                             'host_chain_s': round(float(np.mean(pipe.timers.get('chain_s', [0]))), 4)},
            'hip_calls_ms': per_call,
            'hip_ms_per_pass': per_pass,
            'roofline': roof,
        }
        if not args.no_cpu_baseline and world == 1 and args.cpu_slices > 0:
            log('cpu baseline')
            res['cpu_baseline'] = cpu_baseline(args, vol.vol, heads, args.cpu_slices)
        else:
            res['cpu_baseline'] = None
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
