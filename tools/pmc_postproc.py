"""Post-processing kernels only (no model), for rocprofv3 --pmc passes: two calibration launches with
known byte counts (16-B/lane and 4-B/lane access patterns), then the whole-stack pipeline once.
usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_postproc.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import ENGINE, FILTERS, MATCH, build_inputs  # noqa: E402
from empanada_amd import _hip  # noqa: E402
from empanada_amd.inference import sharded  # noqa: E402

D, S = int(os.environ.get('PMC_DEPTH', 256)), 512
dev = torch.device('cuda', 0)
vol, heads, n_obj = build_inputs(D, S, dev)
torch.cuda.synchronize()
# calibration A: emp_bn_act_nhwc streams float4 (16 B/lane): reads 4 B, writes 4 B per element
x = torch.randn(32, 64, 512, 512, device=dev).contiguous(memory_format=torch.channels_last)   # 2 GiB
sc = torch.ones(64, device=dev)
sh = torch.zeros(64, device=dev)
_hip.bn_act_nhwc_(x, sc, sh, None, True)
# calibration B: harden only (ks = 1): reads 4 B (dword per lane), writes 1 B per voxel
_hip.median_harden_stack(heads['sem'], 1, 0.3)
torch.cuda.synchronize()
for _ in range(2):
    pan = sharded.sharded_panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'], coarse_boundaries=False,
                                         **ENGINE)
    out = sharded.sharded_stack_volume(pan, [1], ENGINE['thing_list'], ENGINE['label_divisor'],
                                       min_size=FILTERS['min_size'], min_span=FILTERS['min_span'], **MATCH)
torch.cuda.synchronize()
print('done', int(out.view(torch.int32).max()))
