"""CPU restatement (numpy) of the device-side tracker / table kernels of empanada_amd/csrc/emp_tracks.hip.

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product.  Each function restates, on plain arrays, what the
reference's InstanceTracker does per slice (empanada/inference/tracker.py:61-123) or what its helpers do, so that the
kernels can be checked one by one on adversarial run tables; the end-to-end equality with the reference's trackers is
pinned separately by the fixtures (tests/golden/pipeline.npz, trackers.npz).
"""
import numpy as np

POS_BITS = 40


def lift_xy_xz(axis, r_start, r_len, r_comp, c_slice, comp_inst, H, W, Y, X, slice0=0, inst_base=0):
    """tracker.py:72-100 for one plane's run table.  Per (instance, slice) the reference has the RLE of the instance's
    pixels (array_utils.py:209-235: runs contiguous in flat 2D index are one run), then maps starts into the volume:
    xy: start + slice * Y * X; xz: (start // W) * Y * X + slice * X + start % W (tracker.py:78-82: lengths kept).
    Returns (key, length) in table order of the run heads."""
    keys, lens = [], []
    n = len(r_start)
    i = 0
    while i < n:
        inst = comp_inst[r_comp[i]]
        if inst < 0:
            i += 1
            continue
        sl = c_slice[r_comp[i]]
        start, length = int(r_start[i]), int(r_len[i])
        j = i + 1
        while (j < n and comp_inst[r_comp[j]] == inst and c_slice[r_comp[j]] == sl
               and int(r_start[j - 1]) + int(r_len[j - 1]) == int(r_start[j])):
            length += int(r_len[j])
            j += 1
        g = int(sl) + slice0
        st3 = start + g * Y * X if axis == 0 else (start // W) * Y * X + g * X + start % W
        keys.append(((inst_base + int(inst)) << POS_BITS) | st3)
        lens.append(length)
        i = j
    return np.array(keys, dtype=np.uint64), np.array(lens, dtype=np.int64)


def lift_yz(vol_inst_plus1, X, x0=0, inst_base=0):
    """tracker.py:83-88 + 110-113: every pixel becomes a unit run at (z, y, slice) and finish() sorts and re-encodes,
    i.e. the RLE along x of the dense labelling.  vol_inst_plus1: (Z, Y, Xl) array of instance + 1 (0 = nothing) of the
    slices [x0, x0 + Xl).  Returns (key, length) of the row runs in raster order (before the touch-merge)."""
    Z, Y, Xl = vol_inst_plus1.shape
    keys, lens = [], []
    for row in range(Z * Y):
        line = vol_inst_plus1.reshape(Z * Y, Xl)[row]
        x = 0
        while x < Xl:
            v = int(line[x])
            if v == 0:
                x += 1
                continue
            x1 = x
            while x1 < Xl and int(line[x1]) == v:
                x1 += 1
            keys.append(((inst_base + v - 1) << POS_BITS) | (row * X + x0 + x))
            lens.append(x1 - x)
            x = x1
    return np.array(keys, dtype=np.uint64), np.array(lens, dtype=np.int64)


def sort_runs(key, length, merge_touching=False):
    """stable sort by key; with merge_touching runs of one instance whose previous end equals their start are joined
    (np.sort + rle_encode of tracker.finish, tracker.py:110-113)."""
    order = np.argsort(key, kind='stable')
    key, length = key[order], length[order]
    if not merge_touching or len(key) == 0:
        return key, length
    ok, ol = [int(key[0])], [int(length[0])]
    for k, l in zip(key[1:].tolist(), length[1:].tolist()):
        if ok[-1] + ol[-1] == k:
            ol[-1] += l
        else:
            ok.append(k)
            ol.append(l)
    return np.array(ok, dtype=np.uint64), np.array(ol, dtype=np.int64)


def offsets(keys_sorted, n_inst):
    """CSR offsets: first run whose instance is >= k, k = 0 .. n_inst"""
    inst = (keys_sorted >> np.uint64(POS_BITS)).astype(np.int64)
    return np.searchsorted(inst, np.arange(n_inst + 1), side='left').astype(np.int64)


def expand(off, obj_val, n_runs):
    return np.repeat(np.asarray(obj_val, dtype=np.int32), np.diff(off))[:n_runs]


def clip(key, length, lo, hi):
    """part of every run inside [lo, hi) (chunk_ranges, empanada/zarr_utils.py:11-47, for one chunk border pair)"""
    mask = np.uint64((1 << POS_BITS) - 1)
    s = (key & mask).astype(np.int64)
    e = s + length
    s2, e2 = np.maximum(s, lo), np.minimum(e, hi)
    keep = s2 < e2
    return (key[keep] & ~mask) | s2[keep].astype(np.uint64), (e2 - s2)[keep]


def reduce_triplets(trip):
    """one (a, b, pixels) row per component pair, pixels summed, ascending (a, b): the sum inside rle_intersection
    (array_utils.py:371-403) over all run pairs of two components"""
    trip = np.asarray(trip, dtype=np.int64).reshape(-1, 3)
    if len(trip) == 0:
        return trip
    key = trip[:, 0] * (1 << 32) + trip[:, 1]
    u, inv = np.unique(key, return_inverse=True)
    v = np.bincount(inv, weights=trip[:, 2]).astype(np.int64)
    return np.stack([u >> 32, u & 0xffffffff, v], axis=1)
