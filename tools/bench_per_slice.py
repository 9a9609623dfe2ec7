"""Throughput of the DROP-IN per-slice protocol -- the calls scripts/pdl_inference3d.py:140-233 makes, one slice at a
time: engine(image) -> pan_seg_to_rle_seg -> apply_matchers, then engine.end(), backward_matching + update_trackers,
finish, filters, fill -- next to the whole-stack protocol bench.py times.  The model forward runs on every slice
(batch 1, timed); its outputs are replaced by the planted heads of that slice so that the post-processing sees a
realistic object load (same convention as bench.py).
`deferred` = the same calls with a deferred engine (empanada_amd/inference/deferred.py): the functions hand handles on
and the stack is evaluated once, in finish_tracking, by the whole-stack path (forward in batches of 16).
`script` = the reference script's PROCESS layout (scripts/pdl_inference3d.py:143-185): the main process runs the engine
(graph) and puts every image, as a numpy array, into an mp.Queue; `forward_matching` runs in a forked mp.Process (which
re-starts itself spawned, patterns._gpu_process_entry) and sends the matched stack back through a Pipe.
usage: PYTHONPATH=. python tools/bench_per_slice.py [n_slices] [plain|tuned|graph|deferred|script|thread] [S]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from empanada_amd.inference import engines as EN
from empanada_amd.inference import filters
from empanada_amd.inference import patterns as PA
from empanada_amd.inference import rle
from empanada_amd.models import GraphedForward, prepare_for_inference, tune_fused_convs


class PlantedModel(torch.nn.Module):
    """runs the real network, hands back the planted head tensors of the current slice"""

    def __init__(self, net, heads):
        super().__init__()
        self.net, self.heads, self.t = net, heads, 0
        self.events = []
        self.checksum = torch.zeros((), dtype=torch.float64, device='cuda')

    def forward(self, x):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = self.net(x.contiguous(memory_format=torch.channels_last))
        e1.record()
        self.events.append((e0, e1))
        self.checksum += out['sem_logits'].float().sum(dtype=torch.float64)
        n = x.size(0)
        t, self.t = self.t, self.t + n
        p = self.heads['sem'][t:t + n].clamp(1e-6, 1 - 1e-6)
        return {'sem_logits': torch.log(p / (1 - p)), 'ctr_hmp': self.heads['ctr_hmp'][t:t + n],
                'offsets': self.heads['offsets'][t:t + n]}


def main():
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    dev = torch.device('cuda')
    vol, heads, n_obj = bench.build_inputs(D, S, dev)
    net = prepare_for_inference(bench.build_model('pdl_r50'), dev)
    mode = sys.argv[2] if len(sys.argv) > 2 else 'plain'            # plain | tuned | graph (= tuned + HIP graph)
    deferred = mode == 'deferred'
    B = 16 if deferred else 1
    if mode in ('tuned', 'graph', 'deferred', 'script', 'thread'):
        x1 = torch.rand((B, 1, S, S), device=dev).contiguous(memory_format=torch.channels_last)
        rep = tune_fused_convs(net, x1)
        counts = {}
        for best, _ in rep.values():
            counts[best] = counts.get(best, 0) + 1
        print(f'conv sites at batch {B}:', counts)
    if mode in ('graph', 'deferred', 'script', 'thread'):
        net = GraphedForward(net)
    model = PlantedModel(net, heads).eval()
    labels, thing, div = [1], bench.ENGINE['thing_list'], bench.ENGINE['label_divisor']
    images = [vol.batch('xy', t, t + 1) for t in range(D)]          # normalised (1,1,H,W) slices, resident
    stages = {}

    def tick(name, t0):
        torch.cuda.synchronize()
        stages[name] = stages.get(name, 0.0) + time.perf_counter() - t0

    for rep in range(2):                                             # pass 0 warms MIOpen / allocator up
        model.t = 0
        model.events = []
        stages.clear()
        eng = EN.PanopticDeepLabEngine3d(model, deferred=deferred, deferred_batch=B, **bench.ENGINE)
        matchers = PA.create_matchers(thing, div, bench.MATCH['merge_iou_thr'], bench.MATCH['merge_ioa_thr'])
        trackers = PA.create_axis_trackers({'xy': 0}, labels, div, (D, S, S))['xy']
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        stack = []
        if mode in ('script', 'thread'):
            import torch.multiprocessing as mp
            if mode == 'thread':                     # the same layout with ONE process on the GPU
                import queue as _q
                import threading
                queue = _q.Queue()
                matcher_out, matcher_in = mp.Pipe()
                proc = threading.Thread(target=PA.forward_matching,
                                        args=(matchers, queue, [], matcher_in, labels, div, thing))
            else:
                queue = mp.Queue()
                matcher_out, matcher_in = mp.Pipe()
                proc = mp.Process(target=PA.forward_matching, args=(matchers, queue, [], matcher_in, labels, div, thing))
            t0 = time.perf_counter()
            proc.start()
            tick('mp.Process(...).start(): fork of the process that holds the GPU', t0)
            if os.environ.get('EMP_SCRIPT_SLEEP'):         # experiment: a pause after the fork
                time.sleep(float(os.environ['EMP_SCRIPT_SLEEP']))
            t0 = time.perf_counter()
            for t in range(D):
                pan = eng(images[t])
                queue.put(None if pan is None else pan.squeeze().cpu().numpy())
            for pan in eng.end():
                queue.put(pan.squeeze().cpu().numpy())
            tick('engine loop (forward + median + pixels + queue.put)', t0)
            t0 = time.perf_counter()
            queue.put('finish')
            stack = matcher_out.recv()[0]
            proc.join()
            tick('wait for the matcher process (its start-up included)', t0)
        for t in range(D if mode not in ('script', 'thread') else 0):
            t0 = time.perf_counter()
            pan = eng(images[t])
            tick('engine (forward + median + pixels)', t0)
            if pan is None:
                continue
            t0 = time.perf_counter()
            seg = rle.pan_seg_to_rle_seg(pan.squeeze().cpu().numpy(), labels, div, thing, True)
            tick('pan_seg_to_rle_seg', t0)
            t0 = time.perf_counter()
            stack.append(PA.apply_matchers(seg, matchers))
            tick('forward matching', t0)
        t0 = time.perf_counter()
        for pan in (eng.end() if mode not in ('script', 'thread') else []):
            seg = rle.pan_seg_to_rle_seg(pan.squeeze().cpu().numpy(), labels, div, thing, True)
            stack.append(PA.apply_matchers(seg, matchers))
        tick('engine.end + tail', t0)
        t0 = time.perf_counter()
        for idx, rs in PA.backward_matching(stack, matchers, len(stack)):
            PA.update_trackers(rs, idx, trackers)
        PA.finish_tracking(trackers)
        for tr in trackers:
            filters.remove_small_objects(tr, min_size=bench.FILTERS['min_size'])
            filters.remove_pancakes(tr, min_span=bench.FILTERS['min_span'])
        tick('backward matching + trackers (deferred: the whole stack is evaluated here) + filters', t0)
        t0 = time.perf_counter()
        out = np.zeros((D, S, S), dtype=np.uint32)
        for tr in trackers:
            PA.fill_volume(out, tr.instances)
        tick('fill', t0)
        dt = time.perf_counter() - t_all
    print(f'per-slice protocol ({mode}), {D}x{S}x{S}: {D * S * S / dt / 1e6:.2f} Mvox/s ({dt / D * 1e3:.2f} ms / slice), '
          f'{len(stack)} slices emitted, {sum(len(t.instances) for t in trackers)} objects')
    fwd = sum(a.elapsed_time(b) for a, b in model.events) * 1e-3
    print(f'  of the engine time, model forward (HIP events): {fwd:.3f} s = {fwd / D * 1e3:.2f} ms / slice')
    for k, v in stages.items():
        print(f'  {k:86s} {v:7.3f} s  {100 * v / dt:5.1f} %')


if __name__ == '__main__':
    main()
