"""Fixture built on the mask the REFERENCE's own post-processing test holds (build container only; same rules and
stand-ins as oracle/gen_golden.py).  `python -m oracle.gen_golden_r3` from the repo root.

  data_post.npz   tests/test_data_post.py:13-43 of the reference: the panoptic ground-truth mask
                  tests/test_data/panoptic/dataset1/masks/pan_seg.tiff (256 x 256; one stuff instance of class 1, seven
                  thing instances of class 2, one stuff instance of class 3) -> training targets (sem class map, centre
                  heat map, centre offsets: PanopticDataset.__getitem__, data/panoptic_dataset.py:72-97, and
                  heatmap_and_offsets, data/utils/target_creation.py:13-78) -> the REFERENCE's
                  get_panoptic_segmentation (inference/postprocess.py:298-356) with the test's arguments.
The dataset classes themselves cannot be imported here (cv2, skimage.io absent), so the targets are restated in this
script with scipy.ndimage.gaussian_filter in the place of cv2.GaussianBlur (49-tap kernel, zero border): they are
INPUTS, stored in the fixture, and the same arrays go to the reference and to this repository's code.  The instance mask
of that test (tests/test_data/instance/dataset1/masks/ins_seg.tiff) is all zeros in the reference tree and is not used.
Fixtures hold DATA only: the mask, the inputs and the reference's outputs.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import REF, _install_standins, _save      # noqa: E402

LABELS, THINGS, DIV = [1, 2, 3], [2], 1000


def targets(mask, sigma=6.0):
    """PanopticDataset.__getitem__ + heatmap_and_offsets for one (h, w) int mask"""
    from scipy import ndimage as ndi
    h, w = mask.shape
    sem = np.zeros_like(mask)
    thing = np.zeros_like(mask)
    for c in LABELS:
        inside = (mask >= c * DIV) & (mask < (c + 1) * DIV)
        sem[inside] = c
        if c in THINGS:
            thing[inside] = mask[inside]
    centers = np.zeros((2, h, w), dtype=np.float32)
    heat = np.zeros((h, w), dtype=np.float32)
    for lab in np.unique(thing):
        if lab == 0:
            continue
        yy, xx = np.nonzero(thing == lab)
        y, x = yy.mean(), xx.mean()                       # regionprops.centroid
        heat[int(y), int(x)] = 1
        centers[0, thing == lab] = y
        centers[1, thing == lab] = x
    heat = ndi.gaussian_filter(heat, sigma, mode='constant', truncate=4.0).astype(np.float32)
    if heat.max() > 0:
        heat = heat / heat.max()
    off = np.zeros_like(centers)
    off[0] = centers[0] - np.arange(h, dtype=np.float32)[:, None]
    off[1] = centers[1] - np.arange(w, dtype=np.float32)[None, :]
    off[:, thing == 0] = 0
    return sem.astype(np.int32), heat[None], off


def main():
    assert os.path.isdir(REF), "reference not mounted: goldens can only be generated in the build container"
    _install_standins()
    import torch
    from PIL import Image
    from empanada.inference import postprocess as PP
    mask = np.array(Image.open(os.path.join(REF, 'tests/test_data/panoptic/dataset1/masks/pan_seg.tiff'))).astype(np.int32)
    sem, heat, off = targets(mask)
    pan, ctr = PP.get_panoptic_segmentation(torch.from_numpy(sem)[None, None].long(), torch.from_numpy(heat)[None],
                                            torch.from_numpy(off)[None], THINGS, DIV, 0, 0, 0.1, 7)
    _save('data_post', mask=mask, sem=sem.astype(np.uint8), ctr_hmp=heat, offsets=off, pan=pan.numpy(), ctr=ctr.numpy())


if __name__ == '__main__':
    main()
