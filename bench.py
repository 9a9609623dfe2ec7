#!/usr/bin/env python
"""bench.py -- end-to-end 3D panoptic inference throughput on MI355X (Mvox/s).

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0).

Default workload (the configuration BASELINE.json's metric is quoted on): ORTHOPLANE inference + consensus on a
1024^3 uint8 volume -- the volume of configs[3], which fits one MI355X; `--size 512` is configs[2].  A "step" is one
full pass of the hot path over the volume (scripts/pdl_inference3d.py:110-233 in orthoplane mode):
  for each plane xy / xz / yz: PanopticDeepLab-R50 forward over every slice (fp32, synthesised weights, resident uint8
  volume) -> sigmoid -> recursive median + harden -> centres -> pixel grouping -> semantic/instance fusion -> runs +
  8-connected components -> slice-to-slice overlaps -> forward/backward label propagation -> 3D trackers -> size/span
  filters;  then instance consensus over the three planes -> filters -> labelled uint32 volume written to a zarr v2
  array chunked (1, Y, X) (scripts/pdl_inference3d.py:225-233).
`--mode stack` keeps round 1's xy-only stack workload (configs[1], 256x512x512, no consensus).

Inputs are resident in HBM when the timed region starts (uint8 EM volume + planted head tensors, see
empanada_amd/synthetic.py and DESIGN.md "Synthetic workload": the conv forward is computed and timed on every slice
of every plane and checksummed, and checked after the timed region against a CPU forward of the same weights; the
post-processing consumes the planted heads so that it sees a realistic object load -- random weights would give it
an empty or degenerate segmentation).

N > 1: `--gpus N` starts N ranks itself (a `torch.distributed.run` child process, before this process touches the
GPU) unless it already runs under one (WORLD_SIZE set: the driver's launch line).  One rank per GPU, RCCL; every plane's
slices are split into contiguous blocks over the ranks of ONE shared volume (strong scaling); see DESIGN.md
"Multi-GPU".
"""
import argparse
import json
import os
import subprocess
import shutil
import sys
import tempfile
import threading
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
CONSENSUS = dict(pixel_vote_thr=2, cluster_iou_thr=0.75, bypass=False)      # scripts/pdl_inference3d.py:44-47
NORM = dict(mean=0.508979, std=0.148561)                      # MitoNet norms
METRIC = 'Mvox/s end-to-end 3D panoptic inference (incl. consensus); PQ vs CPU ref'
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
# HBM traffic / algorithmic bytes from rocprofv3 PMC passes over this script (2 x FETCH_SIZE + WRITE_SIZE, calibrated
# on known byte counts), collected OFFLINE with the tuned implementation choices replayed: --pmc cannot run inside this
# script.  The JSON line labels the figure `traffic_source: offline PMC`.
# emp_conv_bn_act_nhwc (the dominant kernel): the DEFAULT workload's own counters, round 3
# (profiles/r3_pmc_bench_ortho1024.md: 9.075e12 B of traffic over the 4 224 launches of a pass -- tiled kernel, its
# residual-prefetch and 64-wide variants, the weight-stationary 1x1 kernel at 1.00-1.07 -- against 7.672e12 algorithmic).
PMC_TRAFFIC_RATIO = {
    'emp_conv_bn_act_nhwc': 1.18, 'emp_bn_act_nhwc': 1.0, 'emp_dwconv_nhwc': 1.06, 'emp_upsample_bilinear': 1.2,
    'emp_median_harden_stack': 1.02, 'emp_find_centers': 1.42, 'emp_group_pixels': 1.54, 'emp_fuse_apply': 1.0,
    'emp_runs_count': 1.0, 'emp_runs_extract': 1.03,
}
PMC_SOURCE = ('offline PMC ratio x algorithmic bytes (profiles/r3_pmc_bench_ortho1024.md: rocprofv3 --pmc FETCH_SIZE / '
              'WRITE_SIZE passes over this workload at --size 1024; the other dense kernels r2_pmc_bench_ortho512.md, '
              'the per-voxel kernels r3_pmc_postproc_1024.md)')
DENSE_KERNELS = ('emp_bn_act_nhwc', 'emp_dwconv_nhwc', 'emp_upsample_bilinear', 'emp_conv_bn_act_nhwc',
                 'emp_conv_splitk_bn_act_nhwc',
                 'emp_conv_bn_act_proj_nhwc', 'emp_wino_input_transform', 'emp_gemm_nt_batched', 'emp_wino_gemm_fused',
                 'emp_wino_output_transform', 'emp_wino4_input_transform', 'emp_wino4_output_transform',
                 'emp_wino3_input_transform', 'emp_wino3_output_transform',
                 'emp_pointwise_out_nhwc', 'emp_bn_relu_maxpool_nhwc', 'emp_slices_to_input', 'emp_gconv3x3_bn_act_nhwc',
                 'emp_stem_conv7_bn_relu_maxpool', 'emp_logits_to_prob', 'emp_pr_upsample2x', 'emp_pr_topk',
                 'emp_pr_point_sample', 'emp_pr_scatter')
MFMA_KERNELS = ('emp_conv_bn_act_nhwc', 'emp_conv_splitk_bn_act_nhwc', 'emp_conv_bn_act_proj_nhwc', 'emp_gemm_nt_batched', 'emp_wino_gemm_fused',
                'emp_gconv3x3_bn_act_nhwc')
MFMA_F32_PEAK_TFLOPS = 157.3                   # dense fp32 matrix peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0                          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FLOPS_PER_VOXEL_PDL_R50 = 414477.0             # PanopticDeepLab-R50, C = 1, per input pixel
FLOPS_PER_VOXEL = {'pdl_r50': 414477.0, 'mitonet_pr': 560798.0}     # SURVEY 8(d); the others were not counted
COARSE = False                                 # mitonet_pr: instance heads at 1/4 resolution (engines.py:248-275, step 4)
MITO = dict(encoder='resnet50', num_classes=1, stage4_stride=16, decoder_channels=256, low_level_stages=[1],
            low_level_channels_project=[32], atrous_rates=[2, 4, 6], aspp_channels=None, aspp_dropout=0.5,
            ins_decoder=True, ins_ratio=0.5)    # projects/mitonet/configs/mmm_panoptic_deeplab_pointrend.yaml:8-28

# --model name -> label in config.workload (the FLOP count above is only known for the headline model)
MODELS = {'pdl_r50': 'PanopticDeepLab/ResNet-50', 'bifpn_r50': 'PanopticBiFPN/ResNet-50',
          'bifpn_regnety': 'PanopticBiFPN/RegNetY-6.4GF',
          'mitonet_pr': 'PanopticDeepLabPR/ResNet-50 = the MitoNet configuration (instance decoder, PointRend semantic '
                        'head with 2 render steps, 1/4-resolution instance heads -> step-4 grouping)'}
MODEL_ARGS = {'mitonet_pr': (2, False)}        # forward(x, render_steps, interpolate_ins): engines.py:248-256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed passes (default: 3 orthoplane, 10 stack)')
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--mode', default='orthoplane', choices=['orthoplane', 'stack'],
                    help='orthoplane = xy/xz/yz + consensus on a cubic volume of side --size (the metric\'s '
                         'configuration); stack = BASELINE configs[1] (xy only, --depth x --size x --size)')
    ap.add_argument('--size', type=int, default=None, help='default: 1024 (orthoplane) / 512 (stack)')
    ap.add_argument('--depth', type=int, default=256, help='stack mode: slices per rank')
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
    ap.add_argument('--cpu-slices', type=int, default=96, help='stack mode: slices of the workload for the CPU baseline')
    ap.add_argument('--cpu-size', type=int, default=320,
                    help='orthoplane mode: side of the corner sub-volume the CPU baseline (and the ids / PQ check) runs on')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-forward-check', action='store_true')
    ap.add_argument('--no-pipeline', action='store_true', help='run the passes strictly one after the other')
    ap.add_argument('--out', default=None, help='zarr v2 directory the labelled volume is written to (default: a '
                    'scratch store under /dev/shm)')
    ap.add_argument('--keep-out', action='store_true', help='do not delete the output store at the end')
    ap.add_argument('--tile', type=int, default=0,
                    help='stack mode: cut every slice into overlapping tiles of this size (BASELINE configs[4]: "tiled '
                         'overlap stitching"): forward per tile, per-tile post-processing, tile merge on device tables')
    ap.add_argument('--tile-overlap', type=int, default=128)
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel of the forward from Python (no HIP graph)')
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 3 if args.mode == 'orthoplane' else 10
    if args.size is None:
        args.size = 1024 if args.mode == 'orthoplane' else 512
    return args


def maybe_spawn(args):
    """`--gpus N` outside a torch.distributed launch: start the N ranks as a child job and relay its result.  This
    process has not touched the GPU (and never will): it only waits."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ:
        return
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    sys.exit(subprocess.call(cmd, env=env))


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


# ----------------------------------------------------------------------------------------------- model
def build_model(name):
    from empanada_amd.models import PanopticBiFPN, PanopticDeepLab, PanopticDeepLabPR, synthesize_weights
    nc = 1 if len(LABELS) == 1 else len(LABELS) + 1              # binary head, or background + T classes
    if name == 'pdl_r50':
        model = PanopticDeepLab(encoder='resnet50', num_classes=nc)
    elif name == 'mitonet_pr':
        model = PanopticDeepLabPR(**dict(MITO, num_classes=nc))
    else:
        model = PanopticBiFPN(encoder={'bifpn_r50': 'resnet50', 'bifpn_regnety': 'regnety_6p4gf'}[name], num_classes=nc)
    model = synthesize_weights(model)
    with torch.no_grad():                     # synthetic He weights are hot: damp the last layer so that logits come out
        for head in (model.semantic_head, model.ins_center, model.ins_xy):     # O(5), the range a trained model's have
            head.head[1].weight.mul_(0.1)
    return model


class Pipeline:
    def __init__(self, args, device):
        from empanada_amd.models import prepare_for_inference
        self.dtype = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}[args.dtype]
        self.model = prepare_for_inference(build_model(args.model), device, self.dtype)
        self.model_args = MODEL_ARGS.get(args.model, ())
        self.tiler = None
        self.device = device
        self.batch = args.batch
        self.tune_batch = args.tune_batch
        self.timers = {}
        self.tuned = {}
        self.conv_impls = args.conv_impls.split(',') if args.conv_impls else None
        self.post_stream = torch.cuda.Stream(device=device)
        self.dense_profile_left = 0               # forward() calls whose dense-path ABI calls are still event-timed
        # One model call = ~350 kernels behind several hundred Python module calls (~45 ms of host time per call, 1.4 s
        # per 1024-slice plane): with the host also running the chain and the consensus tables, launching is what the
        # pass waits for.  The call is captured once per input shape in a HIP graph and replayed (one launch); the
        # forwards whose kernels are individually event-timed for the roofline block run un-captured.
        self.graphed = None
        # (round 3: a hipMemsetAsync inside the captured PointRend step -- a memset NODE of the graph -- went wrong on
        # replay once other launches had run in between, up to GPU memory faults; tools/diag_mitonet.py, DESIGN.md
        # section 9.  The captured forward contains kernels only now.)
        self.graphs_enabled = True
        if not args.no_graph:
            from empanada_amd.models.graphed import GraphedForward
            self.graphed = GraphedForward(self.model, warmup=1, max_graphs=4, clone_outputs=False)

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
        rep = tune_fused_convs(self.model, x, allow=self.conv_impls, model_args=self.model_args)
        for _, (best, _) in rep.items():
            counts[best] = counts.get(best, 0) + 1
        self.tuned = counts
        saved = sum(t['miopen'] - min(t.values()) for _, t in rep.values())
        log(f'conv call sites tuned: {counts}; isolated saving {saved:.2f} ms per {n} slices')
        if save:
            json.dump({k: v[0] for k, v in rep.items()}, open(save, 'w'), indent=1)

    def replays_next_forward(self):
        """True if the next forward() call replays captured graphs (cheap to queue far ahead)"""
        return (self.graphed is not None and self.graphs_enabled and self.dtype == torch.float32
                and self.dense_profile_left == 0)

    def tune_on_rank0(self, size, rank, save=None, load=None):
        """N ranks: only rank 0 runs the search (MIOpen's find + the per-site timing loop, ~50 s); the others wait for
        its choice per call site and adopt it -- a slice's logits must not depend on the rank that computed it, and
        N - 1 redundant searches whose results are thrown away only lengthen the warm-up."""
        import torch.distributed as dist
        from empanada_amd.models.panoptic_deeplab import FusedConvBNAct
        sites = [(n, m) for n, m in self.model.named_modules() if isinstance(m, FusedConvBNAct)]
        if rank == 0:
            self.tune(size, save, load)
        box = [({n: m.impl for n, m in sites}, self.tuned) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        choice, self.tuned = box[0]
        for n, m in sites:
            m.impl = choice.get(n, m.impl)

    @torch.no_grad()
    def forward(self, dv, axis='xy', lo=0, hi=None):
        """slices [lo, hi) of one plane of the resident uint8 volume (empanada_amd.data.DeviceVolume: strided
        gather + normalise + pad in one HIP pass) -> resident sem probabilities (n,C,h,w) fp32 + a checksum of all
        heads (fp64, device)"""
        from empanada_amd import _hip
        model = self.model
        if self.dense_profile_left > 0:
            self.dense_profile_left -= 1
            _hip.PROFILE_SKIP.difference_update(DENSE_KERNELS)
        else:
            _hip.PROFILE_SKIP.update(DENSE_KERNELS)
            if self.graphed is not None and self.graphs_enabled and self.dtype == torch.float32:
                model = self.graphed
        hi = dv.n_slices(axis) if hi is None else hi
        h, w = dv.plane_shape(axis)
        nc = 1 if len(LABELS) == 1 else len(LABELS) + 1
        prob = torch.empty((hi - lo, nc, h, w), dtype=torch.float32, device=self.device)
        chk = torch.zeros((), dtype=torch.float64, device=self.device)
        per = self.slices_per_call(h, w)
        hp, wp = dv.padded_shape(axis)
        for s in range(lo, hi, per):
            e = min(hi, s + per)
            _hip.trace(f'forward {axis} [{s}, {e}) {"graph" if model is self.graphed else "eager"}')
            buf = model.input_buffer((e - s, 1, hp, wp), args=self.model_args) if model is self.graphed else None
            x = dv.batch(axis, s, e, out=buf)        # straight into the graph's input: no device-to-device copy
            if self.dtype != torch.float32:
                x = x.to(self.dtype)
            out = model(x.contiguous(memory_format=torch.channels_last), *self.model_args)
            logits = out['sem_logits'][..., :h, :w].float()          # logits_to_prob, engines.py:22-30
            dst = prob[s - lo:s - lo + x.shape[0]]                   # written in place: no temporary + device copy
            if self.dtype == torch.float32:
                _hip.logits_to_prob(logits, out=dst)                 # D2 (emp_logits_to_prob)
            elif nc == 1:
                torch.sigmoid(logits, out=dst)
            else:
                torch.softmax(logits, dim=1, out=dst)
            chk += (out['ctr_hmp'].float().sum(dtype=torch.float64) + out['offsets'].float().sum(dtype=torch.float64)
                    + dst.sum(dtype=torch.float64))          # per call: the fp64 temporary stays one batch large
        return prob, chk

    @torch.no_grad()
    def forward_tiled(self, dv, tiler):
        """stack mode with --tile: the model runs on every tile's crop of every slice (inference/tile.py:170-194) --
        (tile area x tiles) / plane area times the pixels of an untiled pass.  Returns the checksum over all tiles."""
        from empanada_amd import _hip
        model = self.model
        if self.dense_profile_left > 0:
            self.dense_profile_left -= 1
            _hip.PROFILE_SKIP.difference_update(DENSE_KERNELS)
        else:
            _hip.PROFILE_SKIP.update(DENSE_KERNELS)
            if self.graphed is not None and self.graphs_enabled and self.dtype == torch.float32:
                model = self.graphed
        chk = torch.zeros((), dtype=torch.float64, device=self.device)
        n = dv.n_slices('xy')
        th, tw = tiler.yranges[0][1] - tiler.yranges[0][0], tiler.xranges[0][1] - tiler.xranges[0][0]
        per = self.slices_per_call(th, tw)
        for s in range(0, n, per):
            x = dv.batch('xy', s, min(n, s + per))
            for (y0, y1), (x0, x1) in zip(tiler.yranges, tiler.xranges):
                xt = x[:, :, y0:y1, x0:x1].contiguous(memory_format=torch.channels_last)
                out = model(xt, *self.model_args)
                prob = _hip.logits_to_prob(out['sem_logits'].float().contiguous())
                chk += (out['ctr_hmp'].float().sum(dtype=torch.float64) + out['offsets'].float().sum(dtype=torch.float64)
                        + prob.sum(dtype=torch.float64))
        return None, chk

    def panoptic(self, heads, tiler=None):
        """planted heads of a stack -> panoptic labels (D, H, W): the whole-plane kernels, or per tile + the tile merge
        (inference/tiled.py; one-run objects, on which the reference's merge raises, are kept)"""
        from empanada_amd.inference import sharded, tiled
        if tiler is None:
            return sharded.sharded_panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'],
                                                  coarse_boundaries=COARSE, **ENGINE)

        def crop(i):
            (y0, y1), (x0, x1) = tiler.yranges[i], tiler.xranges[i]
            return {k: v[:, :, y0:y1, x0:x1].contiguous() for k, v in heads.items()}
        kw = {k: v for k, v in ENGINE.items() if k not in ('thing_list', 'label_divisor')}
        return tiled.tiled_panoptic_stack(crop, heads['sem'].shape[0], tiler, LABELS, thing_list=ENGINE['thing_list'],
                                          label_divisor=ENGINE['label_divisor'], coarse_boundaries=COARSE,
                                          on_single_run='keep', **kw)

    def postprocess(self, heads, out_host):
        """stack mode: probabilities -> labelled slab in pinned host memory.  Same code path for 1 and N ranks
        (empanada_amd/inference/sharded.py); with one rank the collectives are no-ops."""
        from empanada_amd.inference import sharded
        pan = self.panoptic(heads, self.tiler)
        vol = sharded.sharded_stack_volume(pan, LABELS, ENGINE['thing_list'], ENGINE['label_divisor'],
                                           min_size=FILTERS['min_size'], min_span=FILTERS['min_span'], **MATCH)
        out_host.copy_(vol.view(torch.int32), non_blocking=True)
        return vol


@torch.no_grad()
def forward_check(args, pipe, dv, axes=('xy', 'yz')):
    """After the timed region: the GPU forward (tuned hand-written kernels, the path that was timed) against a torch-CPU
    forward of the same synthesised weights on one slice per listed plane; tolerance of the D1 row,
    1e-4 * |ref|_inf + 1e-6 per head: relative to the head's own full scale (tests/conftest.py::dense_tol, the same
    bound the GPU tests use)."""
    cpu_model = build_model(args.model).eval()
    worst, ok, detail = 0.0, True, {}
    for axis in axes:
        z = dv.n_slices(axis) // 2
        x = dv.batch(axis, z, z + 1)
        got = pipe.model(x.contiguous(memory_format=torch.channels_last), *pipe.model_args)
        ref = cpu_model(x.cpu(), *pipe.model_args)
        for k in ('sem_logits', 'ctr_hmp', 'offsets'):
            r = ref[k].float()
            g = got[k].float().cpu()
            err = float((g - r).abs().max())
            tol = 1e-4 * float(r.abs().max()) + 1e-6
            detail[f'{axis}:{k}'] = [round(err, 8), round(tol, 8), round(float(r.abs().max()), 4)]
            ok = ok and err <= tol
            worst = max(worst, err / tol)
    return {'ok': bool(ok), 'worst_err_over_tol': round(worst, 4), 'slices': [f'{a}[{dv.n_slices(a) // 2}]' for a in axes],
            'tolerance': '1e-4*|ref|inf+1e-6 per head', 'max_abs_err_tol_refmax': detail}


# ----------------------------------------------------------------------------------------------- inputs
def build_inputs(D, S, device, seed_offset=0, things=1):
    """stack mode (and tests/test_full_size_gpu.py): (DeviceVolume, planted heads of the xy plane, #objects)"""
    from empanada_amd import synthetic as SY
    from empanada_amd.data import DeviceVolume
    shape = (D, S, S)
    vol = DeviceVolume(SY.em_volume(shape, seed=1234 + seed_offset), NORM['mean'], NORM['std'], 16, device)
    lab, cls = SY.planted_labels(shape, fill=0.08, rmin=6, rmax=24, seed=4321 + seed_offset, n_classes=things)
    lab_dev = torch.from_numpy(lab.view(np.int16)).to(device)
    heads = {'sem': [], 'ctr_hmp': [], 'offsets': []}
    for s in range(0, D, 64):                    # chunked to bound the generator's temporaries
        h = SY.planted_heads(lab_dev, cls, 'xy', device=device, slices=slice(s, min(D, s + 64)), seed=99 + s,
                             n_classes=things, coarse=COARSE)
        for k in heads:
            heads[k].append(h[k])
    heads = {k: torch.cat(v, dim=0).contiguous() for k, v in heads.items()}
    return vol, heads, int(cls.shape[0] - 1)


def shared_host_inputs(shape, rank, things):
    """N ranks of one node: rank 0 draws the synthetic EM volume and the planted label volume ONCE into files under
    /dev/shm, the others map them (the draws are seeded, so this changes nothing but the host time and memory: eight
    ranks each drawing 1024^3 labels on the same host cores would take eight times the CPU for identical arrays)."""
    import torch.distributed as dist
    from empanada_amd import synthetic as SY
    base = '/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir()
    d = os.path.join(base, f'emp_bench_inputs_{os.environ.get("MASTER_PORT", "0")}')
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        lab, cls = SY.planted_labels(shape, fill=0.08, rmin=6, rmax=24, seed=4321, n_classes=things)
        np.save(os.path.join(d, 'lab.npy'), lab)
        np.save(os.path.join(d, 'cls.npy'), cls)
        np.save(os.path.join(d, 'em.npy'), SY.em_volume(shape, seed=1234))
    dist.barrier()
    em = np.load(os.path.join(d, 'em.npy'), mmap_mode='r')
    lab = np.load(os.path.join(d, 'lab.npy'), mmap_mode='r')
    cls = np.load(os.path.join(d, 'cls.npy'))
    em, lab = np.array(em), np.array(lab)            # private copies: the files go away below
    dist.barrier()
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True)
    return em, lab, cls


def build_inputs_ortho(S, device, rank=0, world=1, labels_out=None, things=1):
    """cubic volume shared by all ranks; every rank holds the whole uint8 EM volume (1 GiB at 1024^3; each plane is a
    strided view of it) and the planted heads of its own contiguous block of slices per plane"""
    from empanada_amd import synthetic as SY
    from empanada_amd.data import DeviceVolume
    from empanada_amd.inference.sharded import shard_bounds
    shape = (S, S, S)
    b = shard_bounds(S, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    if world > 1:
        em, lab, cls = shared_host_inputs(shape, rank, things)
        dv = DeviceVolume(em, NORM['mean'], NORM['std'], 16, device)
    else:
        box = {}
        t = threading.Thread(target=lambda: box.update(lab=SY.planted_labels(shape, fill=0.08, rmin=6, rmax=24,
                                                                             seed=4321, n_classes=things)))
        t.start()                                  # numpy on the host, while the EM volume is drawn and uploaded
        dv = DeviceVolume(SY.em_volume(shape, seed=1234), NORM['mean'], NORM['std'], 16, device)
        t.join()
        lab, cls = box['lab']
    if labels_out is not None:
        labels_out.update(lab=lab, cls=cls)
    lab_dev = torch.from_numpy(lab.view(np.int16)).to(device)
    heads, stacks = {}, {}
    chunk = max(1, 64 * 512 * 512 // (S * S))
    for axis in ('xy', 'xz', 'yz'):
        stacks[axis] = (dv, axis, lo, hi)          # the rank's block of the plane: a strided view of the volume
        parts = {'sem': [], 'ctr_hmp': [], 'offsets': []}
        for s in range(lo, hi, chunk):
            h = SY.planted_heads(lab_dev, cls, axis, device=device, slices=slice(s, min(hi, s + chunk)), seed=99 + s,
                                 n_classes=things, coarse=COARSE)
            for k in parts:
                parts[k].append(h[k])
        heads[axis] = {k: torch.cat(v, dim=0).contiguous() for k, v in parts.items()}
    del lab_dev
    return stacks, heads, int(cls.shape[0] - 1), lo


# ----------------------------------------------------------------------------------------------- orthoplane
def postprocess_planes(heads, shape3d, writer, stages, between=None, before=None):
    """The part of a pass that follows the forwards, for all three planes (heads[axis] = the rank's block of head
    tensors): per plane pixels -> tables -> chain (replicated on every rank) -> device-resident trackers, then filters
    -> consensus -> filters -> fill of the rank's z-slab -> pinned host memory -> zarr chunk files.  `between(i)` is called once plane
    i's device tables are on the host (the driver queues the next forward there), `before(axis)` right before plane
    `axis` is touched (the driver makes the post-processing stream wait for that plane's forward there).
    writer: {class: SlabWriter} or None.
    Returns (#consensus instances, {class: the rank's slab of that class's labelled volume on the device}, (z0, z1))."""
    from empanada_amd import _hip
    from empanada_amd.inference import sharded
    planes, base = {}, 0
    for i, axis in enumerate(('xy', 'xz', 'yz')):
        t0 = time.perf_counter()
        if before is not None:
            before(axis)
        h = heads[axis]
        _hip.trace(f'postprocess {axis}')
        pan = sharded.sharded_panoptic_stack(h['sem'], h['ctr_hmp'], h['offsets'], coarse_boundaries=COARSE, **ENGINE)
        table, host = sharded.sharded_tables(pan, LABELS, ENGINE['thing_list'], ENGINE['label_divisor'])
        t1 = time.perf_counter()
        if between is not None:
            between(i)
        t2 = time.perf_counter()
        planes[axis] = sharded.finish_plane(table, host, pan.shape[0], axis, shape3d, LABELS, ENGINE['thing_list'],
                                            ENGINE['label_divisor'], inst_base=base, **MATCH)
        base += planes[axis].n_inst
        stages[f'{axis}_wait_forward_pixels_tables'] = stages.get(f'{axis}_wait_forward_pixels_tables', 0) + t1 - t0
        stages[f'{axis}_enqueue_next_forward'] = stages.get(f'{axis}_enqueue_next_forward', 0) + t2 - t1
        stages[f'{axis}_chain_and_lift'] = stages.get(f'{axis}_chain_and_lift', 0) + time.perf_counter() - t2
    t0 = time.perf_counter()
    cons, vols, zs = sharded.consensus_volume(planes, shape3d, LABELS, ENGINE['thing_list'],
                                              CONSENSUS['pixel_vote_thr'], CONSENSUS['cluster_iou_thr'],
                                              CONSENSUS['bypass'], FILTERS['min_size'], FILTERS['min_span'])
    t1 = time.perf_counter()
    if writer is not None:                             # slab -> pinned host buffer -> chunk files of the zarr array
        for c in LABELS:                               # (written by a thread pool while the GPU runs the next pass)
            writer[c].next_buffer().copy_(vols[c].view(torch.int32), non_blocking=True)
        torch.cuda.current_stream().synchronize()      # the post stream only: a prefetched forward keeps running
        for c in LABELS:
            writer[c].submit()
    stages['consensus_and_fill'] = stages.get('consensus_and_fill', 0) + t1 - t0
    stages['to_host_and_submit_write'] = stages.get('to_host_and_submit_write', 0) + time.perf_counter() - t1
    return sum(int(cons[c].alive.sum()) for c in LABELS), vols, zs


def orthoplane_step(pipe, stacks, heads, shape3d, writer, stages, first=None, prefetch_next=False):
    """One pass.  Two HIP streams: forwards on the default stream, everything downstream of a plane's forward (pixel
    kernels, tables, chain, lift; after the third plane consensus, fill, D2H, zarr write) on the post-processing stream
    behind that forward's event.
    When the forwards replay as HIP graphs (a dozen launches per plane) all of this pass's remaining forwards -- and,
    with prefetch_next, the xy forward of the NEXT pass -- are queued up front: the GPU never waits for the host, which
    matters most when the slices are sharded over several ranks and a plane's forward is as short as its host work.
    Un-captured forwards (~11 000 launches per plane, more than the HIP queue holds) are queued one plane ahead instead:
    plane p+1 as soon as the device tables of plane p are on the host.  The caller hands the returned
    (checksum, event) back as `first`."""
    post = pipe.post_stream
    planes = ('xy', 'xz', 'yz')
    state = {'next': None}
    ev, chk = {}, {}

    def launch(axis, key=None):
        key = key or axis
        with torch.cuda.stream(torch.cuda.default_stream()):
            _, chk[key] = pipe.forward(*stacks[axis])
            ev[key] = torch.cuda.Event()
            ev[key].record()

    if first is None:
        launch('xy')
    else:
        chk['xy'], ev['xy'] = first
    ahead = pipe.replays_next_forward()
    if ahead:
        launch('xz')
        launch('yz')
        if prefetch_next:
            launch('xy', 'next')

    def between(i):
        if ahead:
            return
        if i + 1 < len(planes):
            launch(planes[i + 1])
        elif prefetch_next:
            launch('xy', 'next')

    def before(axis):                                # plane `axis` starts: its forward must have finished
        assert axis in ev, f'forward of plane {axis} was never queued'
        post.wait_event(ev[axis])

    with torch.cuda.stream(post):
        n_found, _, _ = postprocess_planes(heads, shape3d, writer, stages, between, before)
    torch.cuda.current_stream().wait_stream(post)
    if 'next' in ev:
        state['next'] = (chk['next'], ev['next'])
    return chk['xy'] + chk['xz'] + chk['yz'], n_found, state['next']


def cpu_baseline_ortho(args, n, cores):
    """CPU leg of the orthoplane workload on a bounded sample: the n^3 corner sub-volume of the same synthetic volume
    recipe (EM voxels, planted objects clipped to the sub-volume, heads derived from them for all three planes).
    torch-CPU forward over the 3n slices + the oracle chain (CPU restatement of the reference) for every plane +
    oracle consensus + fill; then the HIP path on exactly the same heads -> ids_identical / PQ on the consensus volume.
    kind = 'port'."""
    from empanada_amd import synthetic as SY
    from empanada_amd.evaluation import volume_pq
    from oracle import pipeline as PL
    shape = (n, n, n)
    torch.set_num_threads(cores)
    em = SY.em_volume(shape, seed=1234)
    T = len(LABELS)
    lab, cls = SY.planted_labels(shape, fill=0.08, rmin=6, rmax=24, seed=4321, n_classes=T)
    heads = {a: SY.planted_heads(lab, cls, a, seed=99, n_classes=T, coarse=COARSE) for a in ('xy', 'xz', 'yz')}
    model = build_model(args.model).eval()
    t0 = time.perf_counter()
    for ax, axis in enumerate(('xy', 'xz', 'yz')):
        x = torch.from_numpy(np.ascontiguousarray(np.moveaxis(em, ax, 0))).float().unsqueeze(1)
        x = (x - 255 * NORM['mean']) / (255 * NORM['std'])
        with torch.no_grad():
            for i in range(n):
                out = model(x[i:i + 1], *MODEL_ARGS.get(args.model, ()))
                _ = torch.sigmoid(out['sem_logits']) if T == 1 else torch.softmax(out['sem_logits'], dim=1)
    t_conv = time.perf_counter() - t0
    # oracle/pipeline.py: the per-pixel stages on `cores` processes, matching / tracking / consensus serial as in the
    # reference (one matcher process, scripts/pdl_inference3d.py:143-151)
    np_heads = {a: {k: v.numpy() for k, v in heads[a].items()} for a in heads}
    refs, n_inst, _ = PL.orthoplane_volume(np_heads, shape, dict(ENGINE, coarse_boundaries=COARSE), MATCH, FILTERS,
                                           CONSENSUS, labels=LABELS, workers=cores)
    dt = time.perf_counter() - t0
    # the HIP path on exactly the same heads
    dev_heads = {a: {k: v.cuda().contiguous() for k, v in heads[a].items()} for a in heads}
    _, vols, _ = postprocess_planes(dev_heads, shape, None, {})
    same, pqs, matched = True, [], [0, 0, 0]
    for c in LABELS:
        got = vols[c].view(torch.int32).cpu().numpy().astype(np.uint32)
        pq, n_gt, n_pred, n_match = volume_pq(refs[c], got)
        same = same and bool(np.array_equal(refs[c], got))
        pqs.append(pq)
        matched = [matched[0] + n_gt, matched[1] + n_pred, matched[2] + n_match]
    return {'value': round(float(n) ** 3 / dt / 1e6, 4), 'unit': 'Mvox/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n}^3 corner sub-volume of the same recipe: all three planes ({3 * n} slices of {n}x{n}) + '
                      f'consensus + fill; conv {t_conv:.1f}s of {dt:.1f}s',
            'objects': int(n_inst), 'pq_vs_cpu_ref': round(min(pqs), 6),
            'ids_identical': same, 'instances_cpu_gpu_matched': matched}


def main_orthoplane(args, device, rank, world):
    import torch.distributed as dist
    from empanada_amd import _hip
    S = args.size
    log(f'orthoplane: building inputs {S}^3 (rank {rank}/{world})')
    stacks, heads, n_obj, slice0 = build_inputs_ortho(S, device, rank, world, things=len(LABELS))
    log(f'inputs ready ({n_obj} planted objects)')
    pipe = Pipeline(args, device)
    if not args.no_tune:
        if world > 1:
            pipe.tune_on_rank0(S, rank, args.save_tune, args.load_tune)   # every rank runs the same kernels
        else:
            pipe.tune(S, args.save_tune, args.load_tune)
    shape3d = (S, S, S)
    from empanada_amd.inference.sharded import shard_bounds
    from empanada_amd.zarr_utils import SlabWriter, ZarrV2Group, open_zarr
    zb = shard_bounds(S, world)                      # every rank paints, copies out and WRITES its own z-slab
    out_dir = args.out or os.path.join('/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir(),
                                       f'emp_bench_{os.environ.get("MASTER_PORT", os.getpid())}.zarr')
    names = {c: ('mito_pred' if len(LABELS) == 1 else f'class{c}_pred') for c in LABELS}
    if rank == 0:                                    # scripts/pdl_inference3d.py:228-231: one dataset per class
        grp = ZarrV2Group(out_dir)
        for c in LABELS:
            grp.create_dataset(names[c], shape=shape3d, dtype=np.uint32, overwrite=True, chunks=(1, None, None))
    if world > 1:
        dist.barrier()
    writer = {c: SlabWriter(open_zarr(os.path.join(out_dir, names[c])), int(zb[rank]),
                            (int(zb[rank + 1] - zb[rank]), S, S), torch.int32, threads=4) for c in LABELS}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        orthoplane_step(pipe, stacks, heads, shape3d, writer, {})
        log(f'warmup {i} done')
    barrier()
    _hip.PROFILE = {}
    pipe.dense_profile_left = 3                      # the dense-path calls of the first pass's three forwards are timed
    stages, chks = {}, []
    t0 = time.perf_counter()
    first = None
    for k in range(args.steps):
        chk, n_found, first = orthoplane_step(pipe, stacks, heads, shape3d, writer, stages, first,
                                              prefetch_next=(k + 1 < args.steps) and not args.no_pipeline)
        chks.append(chk)
    for w in writer.values():
        w.drain()                                    # the last pass's chunk files are on disk
    barrier()
    dt = time.perf_counter() - t0
    log(f'timed {args.steps} steps in {dt:.2f}s')
    prof, _hip.PROFILE = _hip.PROFILE, None
    _hip.PROFILE_SKIP.clear()
    per_rank = [dt]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
        allt = torch.zeros((world,), dtype=torch.float64, device=t.device)
        dist.all_gather_into_tensor(allt, t)
        per_rank = [float(v) for v in allt.cpu()]
        dt = max(per_rank)                           # the job is as slow as its slowest rank
    for w in writer.values():
        w.close()
    if rank != 0:
        return
    written = open_zarr(os.path.join(out_dir, names[LABELS[0]]))
    mid = written[S // 2]                            # read one slice back from the store
    out_check = {'path': out_dir, 'datasets': [names[c] for c in LABELS], 'chunks': list(written.chunks),
                 'dtype': str(written.dtype), 'labels_in_slice_read_back': int(len(np.unique(mid)) - 1)}
    chks = [float(c) for c in chks]
    vox_launch = float(S) ** 3 / world               # voxels one post-processing launch covers (a rank's block of a plane)
    if len(LABELS) == 1:
        thing_frac = float((heads['xy']['sem'] >= ENGINE['confidence_thr']).float().mean().item())
    else:
        thing_frac = float((heads['xy']['sem'].argmax(dim=1) > 0).float().mean().item())
    roof, per_call, per_pass = roofline_block(prof, vox_launch, thing_frac, args.steps, dense_passes=1)
    fwd_ms_pass = sum(v for k, v in per_pass.items() if k in DENSE_KERNELS)
    res = {
        'metric': METRIC,
        'value': round(float(S) ** 3 * args.steps / dt / 1e6, 3), 'unit': 'Mvox/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 2),
        'ms_per_step_per_rank': [round(v / args.steps * 1e3, 2) for v in per_rank],
        'world_size': dist.get_world_size() if world > 1 else 1,
        'backend': dist.get_backend() if world > 1 else None,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f32' if args.dtype == 'fp32' else args.dtype, 'data': 'synthetic',
        'config': {'workload': f'orthoplane (xy/xz/yz) inference + instance consensus, {S}^3 uint8 volume '
                               f'(BASELINE configs[{3 if S >= 1024 else 2}] volume), {MODELS[args.model]} '
                               f'C={1 if len(LABELS) == 1 else len(LABELS) + 1} fp32 '
                               f'forward on every slice of every plane + HIP post-processing on planted heads '
                               f'(ks=7, {"1/4-res instance" if COARSE else "full-res"} heads), {n_obj} planted objects, slices of every plane sharded over '
                               f'{world} rank(s); labelled uint32 volume copied to the host and written as a zarr v2 '
                               f'array (chunks (1,Y,X), uncompressed) under {os.path.dirname(out_dir)}'
                               + (' (tmpfs: the write costs memory bandwidth, not disk)' if out_dir.startswith('/dev/shm')
                                  else ''),
                   'mode': 'orthoplane', 'size': S, 'objects_found': int(n_found), 'output': out_check,
                   'batch': pipe.slices_per_call(S, S)},
        'breakdown': {'stages_s_per_step': {k: round(v / args.steps, 4) for k, v in stages.items()},
                      'hand_written_dense_ms_per_pass_rank0': round(fwd_ms_pass, 1),
                      'forward_TFLOPs_if_gpu_bound': round(3 * FLOPS_PER_VOXEL[args.model] * float(S) ** 3 / world
                                                           / (dt / args.steps) / 1e12, 2)
                      if args.model in FLOPS_PER_VOXEL and len(LABELS) == 1 else None,
                      'pipelined': not args.no_pipeline, 'conv_impls': pipe.tuned},
        'forward_checksum': chks[-1], 'forward_checksum_stable': bool(all(c == chks[0] for c in chks)),
        'hip_calls_ms': per_call, 'hip_ms_per_pass': per_pass, 'roofline': roof,
    }
    if not args.no_forward_check:
        log('forward check against the CPU')
        res['forward_check'] = forward_check(args, pipe, stacks['xy'][0])
    if not args.no_cpu_baseline and world == 1 and args.cpu_size > 0:
        log('cpu baseline')
        del heads
        torch.cuda.empty_cache()
        res['cpu_baseline'] = cpu_baseline_ortho(args, min(args.cpu_size, S), min(16, os.cpu_count() or 1))
    else:
        res['cpu_baseline'] = None
    if not args.keep_out and args.out is None:
        shutil.rmtree(out_dir, ignore_errors=True)
    print(json.dumps(res), flush=True)


# ----------------------------------------------------------------------------------------------- accounting
def roofline_block(prof, vox_launch, thing_frac, steps, dense_passes):
    """HIP-event brackets of every ABI call inside the timed region (events on the stream the kernels are launched
    on, empanada_amd/_hip.py:call) -> per-call statistics and the `roofline` object of the dominant hand-written
    kernel.  The per-voxel kernels process a whole block of slices per launch (ALG_BYTES x vox_launch); the dense-path
    kernels report their algorithmic bytes / flops per call (shapes vary by layer) and are sampled during
    `dense_passes` of the timed passes so that event packets do not perturb the rest."""
    stat = {}
    for name, evs in prof.items():
        ms = [e[0].elapsed_time(e[1]) for e in evs]
        by = [e[2] if e[2] is not None else (ALG_BYTES[name](thing_frac) * vox_launch if name in ALG_BYTES else None)
              for e in evs]
        fl = [e[3] for e in evs]
        passes = dense_passes if name in DENSE_KERNELS else steps
        stat[name] = {'calls_per_pass': len(ms) / passes, 'ms_per_pass': float(np.sum(ms)) / passes,
                      'avg_ms': float(np.mean(ms)), 'n': len(ms),
                      'bytes': float(np.sum(by)) if all(b is not None for b in by) else None,
                      'flops': float(np.sum(fl)) if all(f is not None for f in fl) else None}
    per_call = {k: round(v['avg_ms'], 4) for k, v in sorted(stat.items(), key=lambda kv: -kv[1]['avg_ms'])}
    per_pass = {k: round(v['ms_per_pass'], 3) for k, v in sorted(stat.items(), key=lambda kv: -kv[1]['ms_per_pass'])}
    gbps = {k: round(v['bytes'] / (v['avg_ms'] * v['n'] * 1e-3) / 1e9, 1) for k, v in stat.items()
            if v['bytes'] is not None}
    tflops = {k: round(v['flops'] / (v['avg_ms'] * v['n'] * 1e-3) / 1e12, 1) for k, v in stat.items()
              if v['flops'] is not None}
    # dominant hand-written kernel = most GPU time per pass among the calls with a byte / flop model
    dom = max((k for k in stat if stat[k]['bytes'] is not None), key=lambda k: stat[k]['ms_per_pass'])
    n_dom = stat[dom]['n']
    alg = stat[dom]['bytes'] / n_dom                                  # algorithmic bytes per launch (mean)
    traffic = round(PMC_TRAFFIC_RATIO[dom] * alg) if dom in PMC_TRAFFIC_RATIO else None
    if dom in MFMA_KERNELS:
        roof = {'bound': 'mfma', 'kernel': dom, 'achieved': tflops[dom], 'peak': MFMA_F32_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': round(tflops[dom] / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': traffic,
                'alg_flops_per_launch': round(stat[dom]['flops'] / n_dom)}
    else:
        roof = {'bound': 'hbm', 'kernel': dom, 'achieved': gbps[dom], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(gbps[dom] / HBM_PEAK_GBS, 4), 'traffic': traffic}
    roof.update({'traffic_source': PMC_SOURCE if traffic is not None else None,
                 'alg_bytes_per_launch': round(alg), 'avg_launch_ms': round(stat[dom]['avg_ms'], 4),
                 'launches_per_pass': stat[dom]['calls_per_pass'],
                 'ms_per_pass': round(stat[dom]['ms_per_pass'], 3), 'thing_fraction': round(thing_frac, 4),
                 'alg_bytes_per_voxel': {k: round(f(thing_frac), 3) for k, f in ALG_BYTES.items()},
                 'all_kernels_GBps': gbps, 'mfma_kernels_TFLOPs': tflops})
    return roof, per_call, per_pass


# ----------------------------------------------------------------------------------------------- stack mode
def cpu_baseline_stack(args, vol_u8, heads, n_slices, pipe=None):
    """The oracle chain (CPU restatement of the reference) + torch-CPU forward on a bounded sample of the
    same workload: the first n_slices slices.  kind = 'port'.  With --tile the forward runs on every tile's crop and the
    panoptic slices come from the reference's tiled sequence (oracle/pipeline.py::tiled_plane_pans)."""
    from oracle import pipeline as PL
    from oracle import postprocess as OP
    from oracle import rle_ops as OR
    from oracle import rle_seg as OS
    tiler = pipe.tiler if pipe is not None else None
    n = min(n_slices, vol_u8.shape[0])
    cores = min(16, os.cpu_count() or 1)          # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    model = build_model(args.model).eval()
    x = vol_u8[:n].cpu().float().unsqueeze(1)
    x = (x - 255 * NORM['mean']) / (255 * NORM['std'])
    sem, ctr, off = (heads[k][:n].cpu().numpy() for k in ('sem', 'ctr_hmp', 'offsets'))
    t0 = time.perf_counter()
    crops = [(slice(None), slice(None))] if tiler is None else [(slice(*yr), slice(*xr)) for yr, xr in
                                                                  zip(tiler.yranges, tiler.xranges)]
    with torch.no_grad():
        for i in range(n):
            for ys, xs in crops:
                out = model(x[i:i + 1, :, ys, xs], *MODEL_ARGS.get(args.model, ()))
                _ = torch.sigmoid(out['sem_logits']) if len(LABELS) == 1 else torch.softmax(out['sem_logits'], dim=1)
    t_conv = time.perf_counter() - t0
    if tiler is None:
        pans = OP.engine3d_stack([sem[t:t + 1] for t in range(n)], [ctr[t:t + 1] for t in range(n)],
                                 [off[t:t + 1] for t in range(n)], coarse_boundaries=COARSE, render=True, **ENGINE)
        pans = [p.squeeze() for p in pans]
    else:
        pans = PL.tiled_plane_pans(sem, ctr, off, dict(ENGINE, coarse_boundaries=COARSE), tiler.yranges, tiler.xranges,
                                   tiler.overlap_rle, labels=LABELS, single_run='keep')
        pans = [p.astype(np.int64) for p in pans]
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
    pan = pipe.panoptic(sub, tiler) if pipe is not None else sharded.sharded_panoptic_stack(
        sub['sem'], sub['ctr_hmp'], sub['offsets'], coarse_boundaries=COARSE, **ENGINE)
    got = sharded.sharded_stack_volume(pan, LABELS, ENGINE['thing_list'], ENGINE['label_divisor'],
                                       min_size=FILTERS['min_size'], min_span=FILTERS['min_span'], **MATCH)
    got = got.view(torch.int32).cpu().numpy().astype(np.uint32)
    pq, n_gt, n_pred, n_match = volume_pq(out, got)
    return {'value': round(vox / dt / 1e6, 4), 'unit': 'Mvox/s', 'cores': cores, 'kind': 'port',
            'sample': f'first {len(pans)} of {vol_u8.shape[0]} slices ({shape[1]}x{shape[2]}), same heads; '
                      f'conv {t_conv:.1f}s of {dt:.1f}s', 'objects': int(sum(len(t.instances) for t in trs)),
            'pq_vs_cpu_ref': round(pq, 6), 'ids_identical': bool(np.array_equal(out, got)),
            'instances_cpu_gpu_matched': [n_gt, n_pred, n_match]}


def main_stack(args, device, rank, world):
    import torch.distributed as dist
    from empanada_amd import _hip
    from empanada_amd.inference import sharded
    D, S = args.depth, args.size
    log(f'building inputs {D}x{S}x{S}')
    vol, heads, n_obj = build_inputs(D, S, device, seed_offset=rank, things=args.things)
    log(f'inputs ready ({n_obj} planted objects); building model')
    pipe = Pipeline(args, device)
    if args.tile:
        from empanada_amd.inference.tile import Tiler
        pipe.tiler = Tiler((S, S), args.tile, args.tile_overlap)
        log(f'{len(pipe.tiler)} tiles of {args.tile} px, overlap >= {args.tile_overlap}')
    fwd = (lambda v: pipe.forward_tiled(v, pipe.tiler)) if args.tile else pipe.forward
    if not args.no_tune:
        pipe.tune(args.tile or S, args.save_tune, args.load_tune)
    host_out = torch.empty((D, S, S), dtype=torch.int32).pin_memory()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        t_w = time.perf_counter()
        prob, chk = fwd(vol)
        out = pipe.postprocess(heads, host_out)
        torch.cuda.synchronize()
        log(f'warmup {i}: total {time.perf_counter() - t_w:.2f}s')
    barrier()
    _hip.PROFILE = {}
    pipe.dense_profile_left = 1 if not args.no_pipeline else args.steps
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * args.steps)]
    chks = []
    t0 = time.perf_counter()
    if args.no_pipeline:
        for k in range(args.steps):
            ev[3 * k].record()
            prob, chk = fwd(vol)
            chks.append(chk)
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
                pan = pipe.panoptic(heads, pipe.tiler)
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
                prob, chk = fwd(vol)                               # asynchronous: only enqueues
                chks.append(chk)
                ev[3 * k + 1].record()
                fwd_done[k].record()
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
    if rank != 0:
        return
    chks = [float(c) for c in chks]
    vox_total = float(D) * S * S * world * args.steps
    fwd_ms = np.mean([ev[3 * k].elapsed_time(ev[3 * k + 1]) for k in range(args.steps)])
    post_ms = np.mean([ev[3 * k + 1].elapsed_time(ev[3 * k + 2]) for k in range(args.steps)])
    if len(LABELS) == 1:
        thing_frac = float((heads['sem'] >= ENGINE['confidence_thr']).float().mean().item())
    else:
        thing_frac = float((heads['sem'].argmax(dim=1) > 0).float().mean().item())
    roof, per_call, per_pass = roofline_block(prof, float(D) * S * S, thing_frac, args.steps,
                                              dense_passes=1 if not args.no_pipeline else args.steps)
    tile_px = sum((y1 - y0) * (x1 - x0) for (y0, y1), (x0, x1) in zip(pipe.tiler.yranges, pipe.tiler.xranges)) \
        if args.tile else S * S                       # pixels the model sees per slice
    flops = FLOPS_PER_VOXEL.get(args.model, 0.0) * D * tile_px
    res = {
        'metric': 'Mvox/s end-to-end 3D panoptic inference, xy stack only (no consensus); PQ vs CPU ref',
        'value': round(vox_total / dt / 1e6, 3), 'unit': 'Mvox/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 2), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if args.dtype == 'fp32' else args.dtype,
        'data': 'synthetic',
        'config': {'workload': f'stack (xy) inference, {D * world}x{S}x{S} uint8 volume (an independent {D}-slice '
                               f'volume per rank), '
                               + (f'every slice cut into {len(pipe.tiler)} overlapping tiles of {args.tile} px (overlap >= '
                                  f'{args.tile_overlap}): forward and post-processing per tile, tile merge on device '
                                  f'tables, ' if args.tile else '') + f'{MODELS[args.model]} '
                               f'C={1 if len(LABELS) == 1 else len(LABELS) + 1} fp-forward on every slice + HIP '
                               f'post-processing on planted heads (ks=7, {"1/4-res instance" if COARSE else "full-res"} heads), {n_obj} planted objects per rank',
                   'mode': 'stack', 'slices_per_rank': D, 'batch': pipe.slices_per_call(S, S),
                   'objects_found': int(len(np.unique(host_out.numpy())) - 1)},
        'breakdown_ms': {'forward': round(float(fwd_ms), 2), 'forward_end_to_slab_on_host': round(float(post_ms), 2),
                         'forward_TFLOPs': round(flops / (fwd_ms * 1e-3) / 1e12, 2)
                         if args.model in FLOPS_PER_VOXEL and len(LABELS) == 1 else None,
                         'pipelined': not args.no_pipeline, 'conv_impls': pipe.tuned,
                         'host_chain_s': round(float(np.mean(pipe.timers.get('chain_s', [0]))), 4)},
        'forward_checksum': chks[-1], 'forward_checksum_stable': bool(all(c == chks[0] for c in chks)),
        'hip_calls_ms': per_call, 'hip_ms_per_pass': per_pass, 'roofline': roof,
    }
    if not args.no_forward_check:
        res['forward_check'] = forward_check(args, pipe, vol, axes=('xy',))
    if not args.no_cpu_baseline and world == 1 and args.cpu_slices > 0:
        log('cpu baseline')
        res['cpu_baseline'] = cpu_baseline_stack(args, vol.vol, heads, args.cpu_slices, pipe)
    else:
        res['cpu_baseline'] = None
    print(json.dumps(res), flush=True)


def main():
    args = parse()
    maybe_spawn(args)
    if float(os.environ.get('EMP_BENCH_WATCHDOG', 0)) > 0:      # stack traces of every thread, every N seconds: where a
        import faulthandler                                      # rank sits if a multi-rank run ever stops moving
        faulthandler.dump_traceback_later(float(os.environ['EMP_BENCH_WATCHDOG']), repeat=True, file=sys.stderr)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus and rank == 0:
        log(f'--gpus {args.gpus} but the launcher started {world} rank(s): reporting n_gpus = {world}')
    import torch.distributed as dist
    backend = os.environ.get('EMP_BENCH_BACKEND', 'nccl')     # 'gloo': rehearse N ranks on fewer GPUs (not a measurement)
    if backend == 'gloo':
        local = local % max(torch.cuda.device_count(), 1)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    if world > 1:
        dist.init_process_group(backend, device_id=device if backend == 'nccl' else None)
    from empanada_amd import _hip
    _hip.load()
    torch.backends.cudnn.benchmark = True

    if args.things > 1:
        LABELS[:] = list(range(1, args.things + 1))
        ENGINE['thing_list'] = list(LABELS)
    global COARSE
    COARSE = args.model == 'mitonet_pr'
    (main_orthoplane if args.mode == 'orthoplane' else main_stack)(args, device, rank, world)
    if world > 1:
        log(f'rank {rank}: leaving the process group')
        dist.destroy_process_group()
    log(f'rank {rank}: done')


if __name__ == '__main__':
    main()
