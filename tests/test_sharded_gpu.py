"""GPU: the multi-rank paths of empanada_amd/inference/sharded.py on real kernels.

  * two ranks over gloo sharing the one GPU of the box (collectives staged through the host -- the code path is the
    one RCCL runs, only the transport differs): median hand-over, halo tables, replicated chain, block-wise lift,
    all-gather + clip of the runs, z-slab consensus and fill; equal and unequal slice blocks; the stitched slabs must
    equal the volume the REFERENCE produced for the same inputs (tests/golden/pipeline.npz), ids included;
  * the RCCL entry points themselves (`nccl` backend) at world size 1: patterns.all_gather, _all_gather_cat,
    _gather_var, _all_reduce_sum, chain_over_ranks.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden
from empanada_amd import synthetic as SY

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params(g, i):
    C, ks, _, head_seed = (int(x) for x in g[f'p{i}_par'])
    thing = [1] if C == 1 else list(range(1, C))
    labels = [1] if C == 1 else list(range(1, C + 1))
    return C, ks, head_seed, thing, labels


def _ortho(lab, cls, C, ks, head_seed, thing, labels, bounds_of, rank):
    """the orthoplane path of bench.py for one rank: its block of every plane -> its z-slab of every class"""
    from empanada_amd.inference import sharded
    shape = lab.shape
    planes, base = {}, 0
    for name, ax in (('xy', 0), ('xz', 1), ('yz', 2)):
        b = bounds_of(shape[ax])
        lo, hi = int(b[rank]), int(b[rank + 1])
        heads = SY.planted_heads(lab, cls, name, n_classes=C, seed=head_seed, coarse=False)
        h = {k: v[lo:hi].cuda().contiguous() for k, v in heads.items()}
        pan = sharded.sharded_panoptic_stack(h['sem'], h['ctr_hmp'], h['offsets'], thing_list=thing,
                                             label_divisor=1000, stuff_area=16, void_label=0, nms_threshold=0.1,
                                             nms_kernel=7, confidence_thr=0.5, median_kernel_size=ks,
                                             coarse_boundaries=False)
        planes[name] = sharded.track_plane(pan, name, shape, labels, thing, 1000, 0.25, 0.25, inst_base=base)
        base += planes[name].n_inst
    cons, vols, zs = sharded.consensus_volume(planes, shape, labels, thing, 2, 0.75, False, 100, 3)
    return ({c: v.cpu().numpy().astype(np.uint32) for c, v in vols.items()}, zs,
            {c: (r.boxes, r.areas, r.alive) for c, r in cons.items()})


def _worker(rank, world, port, case, split_kind, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        g = load_golden('pipeline')
        C, ks, head_seed, thing, labels = _params(g, case)

        def bounds_of(n):
            from empanada_amd.inference.sharded import shard_bounds
            if split_kind == 'even':
                return shard_bounds(n, world)
            cut = max(ks // 2 + 1, n // 3)               # unequal blocks, each at least as long as the median's reach
            return np.array([0, cut, n]) if world == 2 else np.array([0, cut, cut + (n - cut) // 2, n])
        q.put((rank,) + _ortho(g[f'p{case}_lab'], g[f'p{case}_cls'], C, ks, head_seed, thing, labels, bounds_of, rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case', [0, 1])
@pytest.mark.parametrize('world,split_kind', [(2, 'even'), (2, 'uneven'), (3, 'uneven')])
def test_ranks_over_gloo_reproduce_the_reference_volume(case, world, split_kind):
    g = load_golden('pipeline')
    if case >= int(g['n']):
        pytest.skip('no such fixture')
    C, ks, head_seed, thing, labels = _params(g, case)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, split_kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, vols, zs, cons = q.get(timeout=300)
        got[r] = (vols, zs, cons)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    zcuts = [got[r][1] for r in range(world)]
    assert zcuts[0][0] == 0 and zcuts[-1][1] == g[f'p{case}_lab'].shape[0]
    assert all(zcuts[r][1] == zcuts[r + 1][0] for r in range(world - 1))
    for cid in labels:
        vol = np.concatenate([got[r][0][cid] for r in range(world)], axis=0)
        np.testing.assert_array_equal(vol, g[f'p{case}_vol{cid}'], err_msg=f'class {cid}')
        for r in range(1, world):                        # the instance tables are replicated: identical on every rank
            for a, b in zip(got[0][2][cid], got[r][2][cid]):
                np.testing.assert_array_equal(a, b)


def test_rccl_entry_points_world1():
    """backend 'nccl' (= RCCL) with a single rank: every collective helper of the N-rank path runs on the device
    and returns what the no-process-group shortcut returns"""
    from empanada_amd.inference import patterns as PA
    from empanada_amd.inference import sharded
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        t = torch.arange(24, dtype=torch.float32, device='cuda').reshape(2, 3, 4)
        lst = PA.all_gather(t)
        assert len(lst) == 1 and torch.equal(lst[0], t)
        assert torch.equal(sharded._all_gather_cat(t), t)
        parts = sharded._gather_var(torch.arange(10, dtype=torch.int64, device='cuda').reshape(5, 2))
        assert len(parts) == 1 and parts[0].shape == (5, 2) and parts[0].is_cuda
        parts = sharded._gather_var(torch.arange(7, dtype=torch.int64))        # host table: staged onto the device
        assert torch.equal(parts[0], torch.arange(7, dtype=torch.int64))
        np.testing.assert_array_equal(sharded._all_reduce_sum(np.array([3, 4, 5], dtype=np.int64)), [3, 4, 5])
        x = torch.rand((9, 1, 8, 8), device='cuda')
        from empanada_amd import _hip
        assert torch.equal(sharded.median_handover(x, 5, 0.5), _hip.median_harden_stack(x, 5, 0.5))
        dist.barrier()
    finally:
        dist.destroy_process_group()


class _ListQueue:
    def __init__(self):
        self.items = []

    def put(self, item):
        self.items.append(item)

    def get(self):
        return self.items.pop(0)


class _Sink:
    sent = None

    def send(self, obj):
        self.sent = obj

    def close(self):
        pass


def _worker_multigpu(rank, world, port, case, q):
    """the loop of scripts/inference3d_multigpu.py:351-378 with this package's names: strided slices per rank
    (DistributedEvalSampler), MultiGPUInferenceEngine.get_instance_cells, patterns.all_gather of sem and cells after
    every slice, rank 0 feeding the queue that forward_multigpu consumes"""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from empanada_amd.inference import engines as EN
        from empanada_amd.inference import patterns as PA
        from empanada_amd.sampler import DistributedEvalSampler
        g = load_golden('forward_multigpu')
        C, ks, seed, n_out = (int(x) for x in g[f'c{case}_par'])
        nthing = 1 if C == 1 else C - 1
        lab, cls = SY.planted_labels((9, 56, 64), fill=0.25, rmin=4, rmax=9, seed=seed, n_classes=nthing)
        heads = SY.planted_heads(lab, cls, 'xy', n_classes=nthing, seed=seed)
        labels = [1] if C == 1 else [1, 2]
        eng = EN.MultiGPUInferenceEngine(torch.nn.Identity(), thing_list=[1], label_divisor=1000, nms_kernel=7,
                                         nms_threshold=0.1, confidence_thr=0.5, coarse_boundaries=False)
        queue = _ListQueue()
        n = lab.shape[0]
        mine = list(DistributedEvalSampler(range(n), num_replicas=world, rank=rank))
        rounds = -(-n // world)
        for k in range(rounds):
            z = mine[k] if k < len(mine) else mine[-1]               # ranks without a slice in the last round repeat one
            sem = heads['sem'][z:z + 1].cuda()
            cells = eng.get_instance_cells(heads['ctr_hmp'][z:z + 1].cuda(), heads['offsets'][z:z + 1].cuda())
            sems = PA.all_gather(sem.cpu())                              # gloo moves host tensors
            cells_all = PA.all_gather(cells.float().cpu())
            if rank == 0:
                for r in range(world):
                    if k * world + r < n:                                # global order restored: k * world + r
                        queue.put((sems[r].cuda(), cells_all[r].cuda()))
        if rank == 0:
            queue.put(('finish', 'finish'))
            sink = _Sink()
            PA.forward_multigpu(PA.create_matchers([1], 1000, 0.25, 0.25), queue, [], sink, 0.5, ks, labels, 1000, [1],
                                16, 0)
            stack = sink.sent[0]
            out = [{c: {k2: (v['box'], np.asarray(v['starts']), np.asarray(v['runs'])) for k2, v in rs[c].items()}
                    for c in labels} for rs in stack]
            q.put((rank, out))
        else:
            q.put((rank, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case', [0, 2])
def test_multigpu_script_flow_over_gloo(case):
    """two ranks, strided slices, all_gather after every slice, forward_multigpu on rank 0: the rle_stack equals the one
    the reference's forward_multigpu produced from the same slices in order (tests/golden/forward_multigpu.npz)"""
    from conftest import unpack_rle_seg
    g = load_golden('forward_multigpu')
    C, ks, seed, n_out = (int(x) for x in g[f'c{case}_par'])
    labels = [1] if C == 1 else [1, 2]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_multigpu, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    stack = got[0]
    assert got[1] is None and len(stack) == n_out
    for z, rs in enumerate(stack):
        exp = unpack_rle_seg(g, f'c{case}_z{z}')
        for c in labels:
            e = exp.get(c, {})
            assert list(rs[c].keys()) == list(e.keys())
            for k, (box, st, rn) in rs[c].items():
                assert tuple(int(b) for b in box) == tuple(e[k]['box'])
                np.testing.assert_array_equal(st, e[k]['starts'])
                np.testing.assert_array_equal(rn, e[k]['runs'])
