import os, torch, numpy as np, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dist.init_process_group('nccl')
t = torch.arange(8, dtype=torch.float32, device='cuda').view(2, 4)
out = torch.empty((2, 4), dtype=torch.float32, device='cuda')
post = torch.cuda.Stream()
with torch.cuda.stream(post):
    dist.all_gather_into_tensor(out, t)
    u = torch.zeros((1, 4, 4), dtype=torch.int32, device='cuda')
    o2 = torch.empty_like(u)
    dist.all_gather_into_tensor(o2, u)
    objs = [None]
    dist.all_gather_object(objs, ({'a': np.arange(100000, dtype=np.int64)}, 7))
    res = [[np.arange(5)]]
    dist.broadcast_object_list(res, src=0)
torch.cuda.current_stream().wait_stream(post)
dist.barrier()
x = torch.tensor([1.5], dtype=torch.float64, device='cuda')
dist.all_reduce(x, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert torch.equal(out, t) and objs[0][1] == 7 and len(objs[0][0]['a']) == 100000 and float(x) == 1.5
dist.destroy_process_group()
print('rccl world-1 collectives ok')
