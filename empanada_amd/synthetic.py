"""Synthetic workloads for parity tests and bench.py (SURVEY.md 8(d) recipe).

There are no trained weights and no EM data offline, so the post-processing is fed
*planted* head tensors derived from a label volume of non-overlapping ellipsoids
exactly the way training targets are derived from ground truth
(reference: empanada/data/utils/target_creation.py:13-78 -- centre heatmap with
sigma 6, offsets = centroid - pixel inside objects, 0 outside), plus a noisy
semantic probability so that the median filter and the threshold do real work.

Everything is torch so the same code runs on the host (tests) or on the GPU
(bench, where the tensors are made resident before the timed region).
"""
import math

import numpy as np
import torch

AXES = {'xy': 0, 'xz': 1, 'yz': 2}


def em_volume(shape, seed=1234, threads=8):
    """uint8 EM-like volume: clip(N(129.8, 37.9), 0, 255) -- 255 * MitoNet norms.  Slice z is drawn from its own
    generator default_rng([seed, z]) in fp32, so the volume is the same however many threads fill it."""
    from concurrent.futures import ThreadPoolExecutor
    vol = np.empty(shape, dtype=np.uint8)

    def fill(z):
        rng = np.random.default_rng([seed, z])
        x = rng.standard_normal(size=shape[1:], dtype=np.float32)
        x *= np.float32(37.9)
        x += np.float32(129.8)
        np.clip(x, 0, 255, out=x)
        vol[z] = x.astype(np.uint8)

    with ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(fill, range(shape[0])))
    return vol


def planted_labels(shape, fill=0.08, rmin=6, rmax=24, seed=4321, n_classes=1, max_tries=200000):
    """Label volume (uint16, ids 1..N) of non-overlapping axis-aligned ellipsoids.

    Returns (labels (D,H,W) uint16, classes (N+1,) uint8 with classes[0] = 0).
    Rejection sampling; stops at the target fill fraction.
    """
    rng = np.random.default_rng(seed)
    D, H, W = shape
    lab = np.zeros(shape, dtype=np.uint16)
    classes = [0]
    target = fill * D * H * W
    filled = 0
    tries = 0
    rmax = max(rmin, min(rmax, min(shape) // 2 - 1))
    while filled < target and tries < max_tries and len(classes) < 65535:
        tries += 1
        r = rng.uniform(rmin, rmax, size=3)
        c = rng.uniform(0, 1, size=3) * np.array(shape)
        lo = np.maximum(np.floor(c - r).astype(int), 0)
        hi = np.minimum(np.ceil(c + r).astype(int) + 1, shape)
        if np.any(hi - lo < 2):
            continue
        zz, yy, xx = np.ogrid[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        m = ((zz - c[0]) / r[0]) ** 2 + ((yy - c[1]) / r[1]) ** 2 + ((xx - c[2]) / r[2]) ** 2 <= 1.0
        sub = lab[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
        # keep a one-voxel moat so that distinct objects never touch (8-connectivity)
        # (3x3x3 dilation clipped to the bounding box, separable: one axis at a time)
        dil = m.copy()
        dil[1:] |= m[:-1]
        dil[:-1] |= m[1:]
        t = dil.copy()
        dil[:, 1:] |= t[:, :-1]
        dil[:, :-1] |= t[:, 1:]
        t = dil.copy()
        dil[:, :, 1:] |= t[:, :, :-1]
        dil[:, :, :-1] |= t[:, :, 1:]
        if np.any(sub[dil] != 0):
            continue
        n = len(classes)
        sub[m] = n
        classes.append(int(rng.integers(1, n_classes + 1)))
        filled += int(m.sum())
    return lab, np.array(classes, dtype=np.uint8)


@torch.no_grad()
def planted_heads(labels, classes, axis, *, n_classes=1, sigma=6.0, noise=0.05, seed=99,
                  device='cpu', slices=None, coarse=False):
    """Head tensors for every slice of `labels` along `axis` ('xy'|'xz'|'yz').

    Returns dict of tensors on `device` (slice-major, the layout the engines consume):
      sem      (S, C, H, W) fp32 probabilities; C = 1 (sigmoid-like) if n_classes == 1
               else n_classes + 1 softmax-like rows that sum to 1
      ctr_hmp  (S, 1, h, w) fp32, max of unit Gaussians (sigma) at int(centroid)
      offsets  (S, 2, h, w) fp32, (cy - y, cx - x) inside objects, 0 outside
    With coarse=True the instance heads are produced at 1/4 resolution (h = H/4,
    offsets still in full-resolution pixel units) like the exported PointRend models
    (reference: empanada/inference/engines.py:257-275, step 4).
    `slices` restricts to a python slice of indices along the axis.
    """
    ax = AXES[axis]
    if torch.is_tensor(labels):                    # label volume already resident (bench: uploaded once, int16 bits)
        view = labels.movedim(ax, 0)
        lab = (view if slices is None else view[slices]).to(device).to(torch.int32).contiguous()
        if labels.dtype == torch.int16:            # uint16 ids stored in an int16 tensor
            lab &= 0xFFFF
    else:                                          # only the requested slices are converted
        view = np.moveaxis(labels, ax, 0)
        view = view if slices is None else view[slices]
        lab = torch.as_tensor(np.ascontiguousarray(view).astype(np.int32)).to(device)
    S, H, W = lab.shape
    n = int(classes.shape[0])
    cls = torch.as_tensor(classes.astype(np.int64), device=device)

    # per (slice, id) centroid of the 2D cross-section: integer sums over the object pixels only
    # (float64 atomics on the shared background key are pathologically slow on the GPU)
    zi, yi, xi = torch.nonzero(lab > 0, as_tuple=True)
    key = zi * n + lab[zi, yi, xi].long()
    cnt = torch.zeros(S * n, dtype=torch.int64, device=device).index_add_(0, key, torch.ones_like(key))
    sy = torch.zeros(S * n, dtype=torch.int64, device=device).index_add_(0, key, yi)
    sx = torch.zeros(S * n, dtype=torch.int64, device=device).index_add_(0, key, xi)
    cy = (sy.double() / cnt.clamp(min=1).double()).view(S, n)
    cx = (sx.double() / cnt.clamp(min=1).double()).view(S, n)
    present = (cnt.view(S, n) > 0)
    present[:, 0] = False

    inside = lab > 0
    cyp = torch.gather(cy, 1, lab.view(S, -1).long()).view(S, H, W)
    cxp = torch.gather(cx, 1, lab.view(S, -1).long()).view(S, H, W)
    gy = torch.arange(H, device=device, dtype=torch.float64).view(1, H, 1)
    gx = torch.arange(W, device=device, dtype=torch.float64).view(1, 1, W)
    offy = torch.where(inside, cyp - gy, torch.zeros((), dtype=torch.float64, device=device))
    offx = torch.where(inside, cxp - gx, torch.zeros((), dtype=torch.float64, device=device))
    offsets = torch.stack([offy, offx], dim=1).float()

    # centre heatmap: max over objects of a unit Gaussian at int(centroid)
    hm = torch.zeros(S * H * W, dtype=torch.float32, device=device)
    sidx, oidx = torch.nonzero(present, as_tuple=True)
    if sidx.numel() > 0:
        py = cy[sidx, oidx].long()
        px = cx[sidx, oidx].long()
        rad = int(math.ceil(3 * sigma))
        d = torch.arange(-rad, rad + 1, device=device)
        wy, wx = torch.meshgrid(d, d, indexing='ij')
        g = torch.exp(-(wy.double() ** 2 + wx.double() ** 2) / (2 * sigma * sigma)).float().reshape(-1)
        wy, wx = wy.reshape(-1), wx.reshape(-1)
        chunk = max(1, (1 << 24) // wy.numel())
        for s in range(0, sidx.numel(), chunk):
            yy = py[s:s + chunk, None] + wy[None]
            xx = px[s:s + chunk, None] + wx[None]
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            flat = (sidx[s:s + chunk, None] * H + yy) * W + xx
            hm.scatter_reduce_(0, flat[ok], g[None].expand_as(flat)[ok], reduce='amax')
    ctr = hm.view(S, 1, H, W)

    if coarse:
        assert H % 4 == 0 and W % 4 == 0
        # 1/4-resolution heads sampled at the pixel centres of each 4x4 cell
        ctr = torch.nn.functional.max_pool2d(ctr, 4)
        offsets = offsets[:, :, ::4, ::4].contiguous()

    gen = torch.Generator(device=device).manual_seed(seed + ax)
    if n_classes == 1:
        p = torch.where(inside, 0.9, 0.1).float()
        p = (p + noise * torch.randn(p.shape, generator=gen, device=device)).clamp_(0, 1)
        sem = p.view(S, 1, H, W)
    else:
        cmap = cls[lab.long()]                                      # (S,H,W) class per pixel, 0 = bg
        logits = noise * 10 * torch.randn((S, n_classes + 1, H, W), generator=gen, device=device)
        logits.scatter_add_(1, cmap.view(S, 1, H, W), torch.full((S, 1, H, W), 4.0, device=device))
        sem = torch.softmax(logits, dim=1)
    return {'sem': sem.contiguous(), 'ctr_hmp': ctr.contiguous(), 'offsets': offsets.contiguous()}


def ball(radius):
    """skimage.morphology.ball contract (used by the reference's tests/test_consensus.py:10-17):
    (2r+1)^3 uint8 array, 1 where x^2+y^2+z^2 <= r^2."""
    n = 2 * radius + 1
    z, y, x = np.mgrid[-radius:radius:n * 1j, -radius:radius:n * 1j, -radius:radius:n * 1j]
    return (x * x + y * y + z * z <= radius * radius).astype(np.uint8)
