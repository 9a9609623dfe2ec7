"""Device-resident slice feeder: the reference reads one slice at a time from a zarr / ndarray on the
host (`empanada/data/volume_dataset.py:7-53`), normalises it with albumentations and pads it; here the uint8 volume
lives in HBM (1 GiB at 1024^3) and every plane is a strided view of it -- `emp_slices_to_input` gathers, normalises
and pads a batch of slices in one pass, for any of the three axes, without transposed copies.

Normalisation arithmetic: albumentations.Normalize(mean, std, max_pixel_value=255) computes, in fp32,
(x - mean*255) * (1 / (std*255)).  albumentations is not installed in this image, so this restates its published
formula: parity with the library is unpinned; the kernel is checked bit for bit against a numpy statement of the formula.
"""
import numpy as np
import torch

from . import _hip

__all__ = ['DeviceVolume', 'VolumeDataset', 'normalize_constants', 'AXES']

AXES = {'xy': 0, 'xz': 1, 'yz': 2}


def normalize_constants(mean, std, max_pixel_value=255.0):
    """(mean*255, 1/(std*255)) as fp32, computed like albumentations does (fp32 products, fp32 reciprocal)"""
    m = np.float32(mean) * np.float32(max_pixel_value)
    s = np.float32(std) * np.float32(max_pixel_value)
    return float(m), float(np.reciprocal(s, dtype=np.float32))


class DeviceVolume:
    """A (D, H, W) uint8 volume resident on the GPU, served as normalised, padded (B, 1, hp, wp) fp32 batches along
    any axis.  `len(dv.plane(axis))` slices of shape dv.plane_shape(axis); padding to a multiple of `factor`
    (factor_pad, inference/postprocess.py:25-36)."""

    def __init__(self, volume, mean, std, factor=16, device='cuda'):
        v = torch.as_tensor(np.ascontiguousarray(volume) if isinstance(volume, np.ndarray) else volume)
        assert v.dtype == torch.uint8 and v.dim() == 3, "uint8 (D, H, W) volume expected"
        self.vol = v.to(device).contiguous()
        self.shape = tuple(self.vol.shape)
        self.factor = int(factor)
        self.mean255, self.inv_std255 = normalize_constants(mean, std)

    def n_slices(self, axis):
        return self.shape[AXES[axis]]

    def plane_shape(self, axis):
        d = AXES[axis]
        return tuple(s for i, s in enumerate(self.shape) if i != d)

    def padded_shape(self, axis):
        h, w = self.plane_shape(axis)
        f = self.factor
        return (-(-h // f) * f, -(-w // f) * f)

    def _strides(self, axis):
        D, H, W = self.shape
        st = (H * W, W, 1)
        d = AXES[axis]
        rest = [i for i in range(3) if i != d]
        return st[d], st[rest[0]], st[rest[1]]

    def batch(self, axis, lo, hi, out=None):
        """slices [lo, hi) of the plane -> (hi-lo, 1, hp, wp) fp32 (memory is NCHW == NHWC for one channel); `out`:
        a contiguous tensor of that shape to fill instead of a new one (e.g. a HIP graph's static input)"""
        _hip.require_gpu()
        n = hi - lo
        assert 0 <= lo <= hi <= self.n_slices(axis)
        h, w = self.plane_shape(axis)
        hp, wp = self.padded_shape(axis)
        ss, sr, sc = self._strides(axis)
        if out is None:
            out = torch.empty((n, 1, hp, wp), dtype=torch.float32, device=self.vol.device)
        assert tuple(out.shape) == (n, 1, hp, wp) and out.dtype == torch.float32 and out.is_contiguous()
        _hip.call('emp_slices_to_input', self.vol.data_ptr() + lo * ss, ss, sr, sc, n, h, w, hp, wp, self.mean255,
                  self.inv_std255, out.data_ptr(), _hip.stream(), alg_bytes=n * h * w + 4 * out.numel())
        return out

    def batches(self, axis, batch, lo=0, hi=None):
        hi = self.n_slices(axis) if hi is None else hi
        for s in range(lo, hi, batch):
            yield s, self.batch(axis, s, min(hi, s + batch))


class VolumeDataset:
    """Map-style dataset over the slices of a volume along an axis, the reference's host-side reader
    (empanada/data/volume_dataset.py:7-53): item = {'index', 'image', 'size'} with `tfs(image=...)['image']` applied.
    `array` may be anything `array_utils.take` can slice (numpy, ZarrV2Array).  `scale` > 1 (power-of-two
    down-sampling through cv2.resize in the reference) is not on the hot path and not supported."""

    def __init__(self, array, axis=0, tfs=None, scale=1):
        if scale != 1:
            raise NotImplementedError("VolumeDataset: down-sampling (scale > 1) is outside the hot path")
        self.array, self.axis, self.tfs, self.scale = array, axis, tfs, scale

    def __len__(self):
        return self.array.shape[self.axis]

    def __getitem__(self, idx):
        from .array_utils import take
        image = np.asarray(take(self.array, idx, self.axis))
        h, w = image.shape
        if self.tfs is not None:
            image = self.tfs(image=image)['image']
        return {'index': idx, 'image': image, 'size': (h, w)}
