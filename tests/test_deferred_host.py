"""CPU: the bookkeeping of the deferred per-slice protocol that needs no kernel -- which engine calls hand a slice out
(the counting rule of the reference's median queue, engines.py:68-90), how handles chain and how a model that cannot be
batched is refused."""
import numpy as np
import pytest
import torch

from empanada_amd.inference import deferred as DF
from empanada_amd.inference import engines as EN


class _Queue(EN._MedianQueue):
    def get_median(self, key):                               # no kernel: the middle item's own value
        return self.median_queue[self.mid_idx][key]


@pytest.mark.parametrize('ks', [1, 3, 5, 7, 11])
def test_deferred_session_hands_slices_out_like_the_median_queue(ks):
    for D in range(0, 2 * ks + 3):
        q = _Queue(ks)
        ref = []
        for t in range(D):
            q.enqueue({'sem': t})
            out = q.get_next(keys=['sem'])
            ref.append(None if out is None else out['sem'])
        tail = [item['sem'] for item in q.end()]

        class Eng:
            pass
        eng = Eng()
        eng.ks, eng.mid_idx = ks, ks // 2
        s = DF.StackSession(eng, batch_size=10 ** 6)         # never flushes: no model involved
        got = [s.add(torch.zeros(1, 1, 4, 4)) for _ in range(D)]
        got_tail = s.end()
        assert [g is None for g in got] == [r is None for r in ref], (ks, D)
        assert len(got_tail) == len(tail)
        # ordinals enumerate the emitted slices in emission order
        ordinals = [g._k for g in got if g is not None] + [g._k for g in got_tail]
        assert ordinals == list(range(len(ordinals)))
        emitted_slices = [r for r in ref if r is not None] + tail
        assert len(emitted_slices) == len(ordinals)
        assert s.closed and s.n_emitted == len(ordinals)


def test_handles_chain_without_computing_and_unknown_uses_compute():
    class Eng:
        ks, mid_idx = 1, 0
    s = DF.StackSession(Eng(), batch_size=10 ** 6)
    pan = s.add(torch.zeros(1, 1, 4, 4))
    chained = pan.squeeze().cpu().numpy()
    assert isinstance(chained, DF.LazyPan) and chained._is2d() and not pan._is2d() and chained._val is None
    s.force_pan = lambda k: torch.arange(16).reshape(1, 1, 4, 4)           # stand-in for the per-slice engine code
    assert np.asarray(chained).shape == (4, 4) and isinstance(chained._val, np.ndarray)
    assert tuple(pan.shape) == (1, 1, 4, 4)                                # attribute access computes
    assert int(torch.sum(pan)) == 120                                      # torch functions compute
    assert (pan + 1)[0, 0, 0, 0] == 1
    assert torch.stack([pan, pan]).shape == (2, 1, 1, 4, 4)                # handles inside containers too


def test_a_model_that_answers_one_image_per_call_is_refused_when_batched():
    class Eng:
        ks, mid_idx = 1, 0

        def _deferred_infer(self, x, upsampling):
            z = torch.zeros(1, 1, 4, 4)
            return {'sem': z, 'ctr_hmp': z, 'offsets': torch.zeros(1, 2, 4, 4)}
    s = DF.StackSession(Eng(), batch_size=2)
    s.add(torch.zeros(1, 1, 4, 4))
    with pytest.raises(RuntimeError, match='deferred_batch=1'):
        s.add(torch.zeros(1, 1, 4, 4))
