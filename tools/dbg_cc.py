import numpy as np, torch, sys
sys.path.insert(0, '.')
from empanada_amd import _hip as hip
from empanada_amd.inference import rle
from oracle import rle_seg as OS
shape = (2, 200, 260)
rng = np.random.default_rng(sum(shape))
D, H, W = shape
base = rng.integers(0, 5, (D, H // 3 + 1, W // 2 + 1))
pan = np.repeat(np.repeat(base, 3, 1), 2, 2)[:, :H, :W]
noise = rng.random((D, H, W)) < 0.15
pan = np.where(noise, rng.integers(0, 5, (D, H, W)), pan)
cls = np.repeat(np.repeat(rng.integers(1, 4, (D, H // 8 + 1, W // 8 + 1)), 8, 1), 8, 2)[:, :H, :W]
pan = np.where(pan > 0, cls * 1000 + pan, 0).astype(np.int64)
pan[:, :, -1] = pan[:, :, 0]
segs, table = rle.stack_to_rle_segs(hip.np_to_dev_u32(pan), [1, 2, 3], 1000, [1, 2], True)
print('n_runs', table.n_runs, 'n_comp', table.n_comp)
for d in range(D):
    exp = OS.pan_seg_to_rle_seg(pan[d], [1, 2, 3], 1000, [1, 2], True)
    for c in (1, 2, 3):
        a, b = segs[d][c], exp[c]
        print(d, c, len(a), len(b), list(a.keys())[:5], list(b.keys())[:5], list(a.keys())[-3:], list(b.keys())[-3:])
        if list(a.keys()) != list(b.keys()):
            ka, kb = list(a.keys()), list(b.keys())
            for i in range(min(len(ka), len(kb))):
                if ka[i] != kb[i]:
                    print(' first key diff at', i, ka[i], kb[i]); break
        # compare as sets of (starts tuple)
        sa = {tuple(v['starts'][:3]) for v in a.values()}; sb = {tuple(v['starts'][:3]) for v in b.values()}
        print('  only gpu', len(sa - sb), 'only oracle', len(sb - sa))
