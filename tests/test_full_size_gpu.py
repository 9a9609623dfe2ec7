"""GPU: BASELINE.json configs[1] at its full size (256 x 512 x 512 stack, ks = 7, full-resolution heads) through the
post-processing path, checked with size-independent properties (the oracle finishes only a sub-stack in seconds, and is
compared there):

  * run-table round trip: painting the run table of the panoptic stack with each run's own value reproduces the
    thing voxels of the stack exactly (encode -> decode);
  * conservation: component areas == run lengths == voxel counts, per slice (a checksum of checksums);
  * the label-propagation chain only merges: every final object is a union of components, its voxel count is the sum
    of their areas, and objects of the planted ground truth are recovered after the size / span filters (PQ >= 0.93 against the planted labels, >= 97 % matched);
  * determinism: a second pass is bit-identical;
  * a sub-stack (first 24 slices) equals the oracle bit for bit, ids included;
  * two virtual ranks with a one-slice halo (the multi-GPU decomposition) give the single-rank volume.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

D, S = 256, 512


@pytest.fixture(scope='module')
def stack():
    import bench
    vol, heads, n_obj = bench.build_inputs(D, S, torch.device('cuda'))
    del vol
    from empanada_amd.inference import sharded
    pan = sharded.sharded_panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'], coarse_boundaries=False,
                                         **bench.ENGINE)
    return bench, heads, pan, n_obj


def test_run_table_round_trip_and_conservation(stack):
    bench, heads, pan, _ = stack
    from empanada_amd import _hip
    from empanada_amd.inference import patterns as PA
    div = bench.ENGINE['label_divisor']
    table, host = PA.tables_from_stack(pan, [1], [1], div)
    assert table.n_comp == len(host['c_area']) > 10000
    # encode -> decode: paint every component with the original value of its runs
    r_val = table.r_val.cpu().numpy()
    comp_val = r_val[table.c_first.cpu().numpy()].astype(np.int64)
    vol = torch.zeros(tuple(pan.shape), dtype=torch.int32, device='cuda').view(torch.uint32)
    _hip.fill_table_u32(vol, table, _hip.np_to_dev_u32(comp_val), slice0=0)
    assert torch.equal(vol.view(torch.int32), pan.view(torch.int32))
    # conservation, per slice: sum of component areas == number of labelled voxels
    per_slice_vox = (pan.view(torch.int32) != 0).sum(dim=(1, 2)).cpu().numpy()
    per_slice_area = np.bincount(host['c_slice'], weights=host['c_area'], minlength=D).astype(np.int64)
    np.testing.assert_array_equal(per_slice_area, per_slice_vox)
    run_len = table.r_len.cpu().numpy().astype(np.int64)
    run_comp = table.r_comp.cpu().numpy()
    np.testing.assert_array_equal(np.bincount(run_comp, weights=run_len, minlength=table.n_comp).astype(np.int64),
                                  host['c_area'])


def test_chain_volume_properties_and_determinism(stack):
    bench, heads, pan, n_obj = stack
    from empanada_amd import synthetic as SY
    from empanada_amd.evaluation import volume_pq
    from empanada_amd.inference import patterns as PA
    from empanada_amd.inference import sharded
    div = bench.ENGINE['label_divisor']
    table, host = PA.tables_from_stack(pan, [1], [1], div)
    final, _ = PA.chain_from_tables(host, D, [1], [1], div, **bench.MATCH)
    assert np.all(final > div) and np.all(final < 2 * div)
    vol = sharded.fill_slab(table, final, tuple(pan.shape))
    ids, counts = torch.unique(vol.view(torch.int32), return_counts=True)
    ids, counts = ids.cpu().numpy(), counts.cpu().numpy()
    areas = np.bincount(final - div, weights=host['c_area']).astype(np.int64)
    for i, c in zip(ids[1:], counts[1:]):
        assert areas[i - div] == c                                     # an object = the union of its components
    assert (vol.view(torch.int32) != 0).eq(pan.view(torch.int32) != 0).all()
    # planted ground truth is recovered
    lab, _ = SY.planted_labels((D, S, S), fill=0.08, rmin=6, rmax=24, seed=4321)
    kept = sharded.filter_labels(host, final, bench.FILTERS['min_size'], bench.FILTERS['min_span'])
    vol_f = sharded.fill_slab(table, kept, tuple(pan.shape))
    pq, n_gt, n_pred, n_match = volume_pq(torch.from_numpy(lab.astype(np.int64)).cuda(), vol_f.view(torch.int32).long())
    assert n_gt == n_obj and pq >= 0.93 and n_match >= 0.97 * n_gt and n_pred <= n_gt, (pq, n_gt, n_pred, n_match)
    # second pass: bit-identical
    pan2 = sharded.sharded_panoptic_stack(heads['sem'], heads['ctr_hmp'], heads['offsets'], coarse_boundaries=False,
                                          **bench.ENGINE)
    assert torch.equal(pan2.view(torch.int32), pan.view(torch.int32))
    vol2 = sharded.sharded_stack_volume(pan2, [1], [1], div, **bench.MATCH)
    assert torch.equal(vol2.view(torch.int32), vol.view(torch.int32))


def test_substack_equals_oracle(stack):
    bench, heads, pan, _ = stack
    from oracle import postprocess as OP
    n = 24
    sub = {k: v[:n].cpu().numpy() for k, v in heads.items()}
    pans = OP.engine3d_stack([sub['sem'][t:t + 1] for t in range(n)], [sub['ctr_hmp'][t:t + 1] for t in range(n)],
                             [sub['offsets'][t:t + 1] for t in range(n)], coarse_boundaries=False, render=True,
                             sizes=[(S, S)] * n, **bench.ENGINE)
    m = bench.ENGINE['median_kernel_size'] // 2
    # the recursive median looks m slices ahead: the first n - m slices of the sub-stack are those of the full stack
    got = pan[:n - m].cpu().numpy().astype(np.int64)
    exp = np.stack([np.asarray(p).squeeze() for p in pans[:n - m]]).astype(np.int64)
    np.testing.assert_array_equal(got, exp)


def test_two_virtual_ranks_equal_single_rank(stack):
    bench, heads, pan, _ = stack
    from empanada_amd.inference import patterns as PA
    from empanada_amd.inference import sharded
    div = bench.ENGINE['label_divisor']
    single = sharded.sharded_stack_volume(pan, [1], [1], div, min_size=bench.FILTERS['min_size'],
                                          min_span=bench.FILTERS['min_span'], **bench.MATCH)
    cut = 100
    tabs, hosts = [], []
    for lo, hi in ((0, cut + 1), (cut, D)):            # rank 0 carries the first slice of rank 1 as halo
        t, h = PA.tables_from_stack(pan[lo:hi].contiguous(), [1], [1], div)
        tabs.append(t)
        hosts.append(h)
    counts = np.array([cut, D - cut])
    merged, own = sharded.merge_rank_tables(hosts, counts)
    final, _ = PA.chain_from_tables(merged, D, [1], [1], div, **bench.MATCH)
    final = sharded.filter_labels(merged, final, bench.FILTERS['min_size'], bench.FILTERS['min_span'])
    parts = []
    for r, (lo, hi) in enumerate(((0, cut), (cut, D))):
        fl = np.zeros(len(own[r]), dtype=np.int64)
        sel = own[r] >= 0
        fl[sel] = final[own[r][sel]]
        slab = sharded.fill_slab(tabs[r], fl, (hi - lo + (1 if r == 0 else 0), S, S))
        parts.append(slab[:hi - lo])
    both = torch.cat([p.view(torch.int32) for p in parts], dim=0)
    assert torch.equal(both, single.view(torch.int32))
