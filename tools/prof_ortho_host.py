"""cProfile of the host side of one orthoplane pass (512^3): where the tail (tracking, consensus) goes.
usage: PYTHONPATH=. python tools/prof_ortho_host.py"""
import cProfile
import pstats
import sys

import torch

sys.argv = ['bench.py', '--mode', 'orthoplane', '--size', '512', '--no-tune']
import bench

args = bench.parse()
device = torch.device('cuda', 0)
from empanada_amd import _hip
_hip.load()
torch.backends.cudnn.benchmark = True
S = args.size
stacks, heads, n_obj, slice0 = bench.build_inputs_ortho(S, device)
pipe = bench.Pipeline(args, device)
host_out = torch.empty((S, S, S), dtype=torch.int32).pin_memory()
bench.orthoplane_step(pipe, stacks, heads, slice0, (S, S, S), host_out, {})
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
stages = {}
bench.orthoplane_step(pipe, stacks, heads, slice0, (S, S, S), host_out, stages)
torch.cuda.synchronize()
pr.disable()
print(stages)
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
