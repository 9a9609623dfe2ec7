"""HIP-graph replay of an inference model's forward, for the launch-bound case.

The whole-stack protocol pushes 128 slices through the model per call: ~350 kernels of ~0.5 ms each, the queue never
runs dry and a graph would buy nothing.  The reference's own scripts, however, call ``engine(image)`` once per slice
(scripts/pdl_inference3d.py:140-160): at batch 1 the forward is ~200 launches of 5-20 us behind several hundred
Python module calls, and the GPU idles between them.  ``GraphedForward`` captures the forward once per input shape
(hipStreamBeginCapture through ``torch.cuda.graph``; the hand-written kernels are launched on torch's current stream,
so they are captured like torch's own) and replays it with a single launch afterwards.

Outputs are cloned out of the graph's static buffers, because the engines keep the head tensors of the last
``median_kernel_size`` slices in their queue.

Caveat found in round 3: on ROCm 7.2 a MEMSET node inside a captured graph goes wrong on replay
once other kernels have been launched between replays (wrong results, then GPU memory faults).  This package's own
capturable entries zero with kernels for that reason; library ops that call ``hipMemsetAsync`` internally
(``torch.topk`` on large rows does) must not be captured -- the PointRend models are safe on the fp32 GPU path (D10
kernels) and must NOT be wrapped in ``GraphedForward`` on the library path (bf16 / fp16).
"""
import torch

__all__ = ['GraphedForward']


class GraphedForward(torch.nn.Module):
    """``model = GraphedForward(prepare_for_inference(model))``; call it like the model.  One graph per distinct
    (input shape, dtype, extra positional / keyword arguments); inputs must live on the model's GPU."""

    def __init__(self, model, warmup=3, max_graphs=8, clone_outputs=True):
        super().__init__()
        self.model = model.eval()
        self.warmup = int(warmup)
        self.max_graphs = int(max_graphs)
        # clone_outputs=False hands out the graph's static output buffers: valid until the next replay of the same
        # graph, which is enough for a consumer that reads them on the same stream before calling again (bench.py)
        self.clone_outputs = bool(clone_outputs)
        self._graphs = {}

    def parameters(self, recurse=True):              # engines look the device up through the first parameter
        return self.model.parameters(recurse)

    @torch.no_grad()
    def _capture(self, x, args, kwargs):
        static_in = x.clone()
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                # warm-up off the capture: MIOpen find, filter transforms, tile tables
            for _ in range(self.warmup):
                self.model(static_in, *args, **kwargs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of OTHER threads (the RCCL watchdog's event queries in a multi-rank job, a writer
        # pool) must not invalidate the capture running on this thread
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            static_out = self.model(static_in, *args, **kwargs)
        return graph, static_in, static_out

    def input_buffer(self, shape, dtype=torch.float32, channels_last=True, args=()):
        """the static input tensor of the graph captured for this input signature (and these extra positional
        arguments), or None: writing the next input straight into it saves the device-to-device copy in forward()"""
        entry = self._graphs.get((tuple(shape), dtype, channels_last, tuple(args), ()))
        return None if entry is None else entry[1]

    @torch.no_grad()
    def forward(self, x, *args, **kwargs):
        if not x.is_cuda:
            raise RuntimeError("GraphedForward needs its input on the GPU")
        key = (tuple(x.shape), x.dtype, x.is_contiguous(memory_format=torch.channels_last), args,
               tuple(sorted(kwargs.items())))
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            entry = self._graphs[key] = self._capture(x, args, kwargs)
        graph, static_in, static_out = entry
        if x.data_ptr() != static_in.data_ptr():      # a caller that filled input_buffer() in place skips this copy
            static_in.copy_(x)
        graph.replay()
        if not self.clone_outputs:
            return dict(static_out)
        return {k: v.clone() for k, v in static_out.items()}
