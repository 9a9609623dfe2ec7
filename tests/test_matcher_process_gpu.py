"""GPU: the reference script's process layout (scripts/pdl_inference3d.py:143-185) -- the main process holds the GPU and
FORKS `forward_matching`, feeds it panoptic images through an mp.Queue and receives the matched stack through a Pipe.  The
forked child cannot use HIP; `forward_matching` re-starts itself in a spawned process (patterns._gpu_process_entry) and the
stack that comes back equals the one computed in-process."""
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_forward_matching_in_a_forked_matcher_process():
    from empanada_amd.inference import patterns as PA
    from empanada_amd.inference import rle
    g = load_golden('pipeline')
    pans = g['p0_xy_pan'].astype(np.int64)
    labels, thing = [1], [1]
    torch.zeros(1).cuda()                                   # the parent has initialised the GPU, like the script
    # in-process result
    matchers = PA.create_matchers(thing, 1000, 0.25, 0.25)
    exp = [PA.apply_matchers(rle.pan_seg_to_rle_seg(p, labels, 1000, thing, True), matchers) for p in pans]
    # the script's way: default start method (fork), numpy images through a queue, None for the slices the median queue
    # swallowed, 'finish' at the end
    matchers = PA.create_matchers(thing, 1000, 0.25, 0.25)
    queue = mp.Queue()
    matcher_out, matcher_in = mp.Pipe()
    proc = mp.Process(target=PA.forward_matching, args=(matchers, queue, [], matcher_in, labels, 1000, thing))
    proc.start()
    queue.put(None)
    for p in pans:
        queue.put(p)
    queue.put('finish')
    assert matcher_out.poll(120), "no answer from the matcher process"
    got = matcher_out.recv()[0]
    proc.join(60)
    assert proc.exitcode == 0
    assert len(got) == len(exp)
    for a, b in zip(got, exp):
        assert list(a.keys()) == list(b.keys())
        for c in a:
            assert list(a[c].keys()) == list(b[c].keys())
            for k in a[c]:
                assert tuple(a[c][k]['box']) == tuple(b[c][k]['box'])
                np.testing.assert_array_equal(a[c][k]['starts'], b[c][k]['starts'])
                np.testing.assert_array_equal(a[c][k]['runs'], b[c][k]['runs'])
    fwd = np.stack([rle.rle_seg_to_pan_seg(rs, pans[0].shape) for rs in got])
    np.testing.assert_array_equal(fwd, g['p0_xy_fwd'])       # and equals the reference's forward-matched stack
