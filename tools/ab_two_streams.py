"""Do two model calls on two HIP streams overlap usefully?  A quarter of a forward pass is memory-bound kernels (Winograd
transforms, depthwise, layer1 1x1, up-sampling, stem), the rest sits on the matrix cores: two calls in different phases
could fill each other's idle unit.  Two captured graphs of the same tuned model (own static buffers), N replays each:
one after the other on one stream vs. side by side on two.
usage: python tools/ab_two_streams.py [slices_per_call] [size] [tune.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from empanada_amd.models import GraphedForward, prepare_for_inference, tune_fused_convs
from empanada_amd.models.panoptic_deeplab import FusedConvBNAct

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device('cuda')
net = prepare_for_inference(bench.build_model('pdl_r50'), dev)
x = [torch.rand((B, 1, S, S), device=dev).contiguous(memory_format=torch.channels_last) for _ in range(2)]
if len(sys.argv) > 3:
    choice = json.load(open(sys.argv[3]))
    for name, m in net.named_modules():
        if isinstance(m, FusedConvBNAct):
            m.impl = choice.get(name, 'miopen')
else:
    tune_fused_convs(net, x[0][:max(1, B // 4)].contiguous(memory_format=torch.channels_last))
g = [GraphedForward(net, warmup=1, clone_outputs=False) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
with torch.no_grad():
    for i in range(2):
        g[i](x[i])
    torch.cuda.synchronize()
    N = 6

    def run(two):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(N):
            for i in range(2):
                with torch.cuda.stream(streams[i if two else 0]):
                    if two and i == 1 and k == 0:
                        pass
                    g[i](x[i])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (2 * N) * 1e3

    for rep in range(2):
        a = run(False)
        b = run(True)
        print(f'{B} x {S}^2 per call: one stream {a:.2f} ms / call, two streams {b:.2f} ms / call ({a / b:.3f}x)')
