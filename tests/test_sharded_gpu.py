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
