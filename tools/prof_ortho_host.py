"""cProfile of the host side of the post-processing of one orthoplane pass (planted heads, no forward): where the
replicated host terms of the N-rank path go (chain, instance tables, consensus tables).
usage: python tools/prof_ortho_host.py [size]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device('cuda', 0)
stacks, heads, n_obj, _ = bench.build_inputs_ortho(S, dev)
del stacks
bench.postprocess_planes(heads, (S, S, S), None, {})
torch.cuda.synchronize()
pr = cProfile.Profile()
stages = {}
pr.enable()
bench.postprocess_planes(heads, (S, S, S), None, stages)
torch.cuda.synchronize()
pr.disable()
print({k: round(v, 4) for k, v in stages.items()})
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
