"""2D tiling of planes that are too large for one forward pass, reference names
(``empanada/inference/tile.py``): ``Tiler`` :54-194 (``translate_rle_seg`` :122-168, ``__call__`` :170-194,
``overlap_mask`` :112-120), ``calculate_overlap_rle`` :8-52.

The reference delegates the tile geometry to the third-party ``cztile`` package
(``AlmostEqualBorderFixedTotalAreaStrategy2D``), which is not available here and is unpinned in the reference
(no test checks tile positions).  The geometry used instead is stated here and pinned by this
repository's own fixtures: tiles have the fixed total size (th, tw) (clipped to the image); along each axis the
number of tiles is the smallest n with n*t - (n-1)*overlap >= L, and the n offsets are spread as evenly as
integer arithmetic allows between 0 and L - t, so that all overlaps are >= overlap_width and almost equal.
Tiles are enumerated row-major (y outer, x inner).
"""
import numpy as np

from ..array_utils import merge_rles, rle_voting

__all__ = ['Tiler', 'calculate_overlap_rle', 'axis_offsets']


def axis_offsets(length, tile, overlap):
    """Start offsets of the tiles along one axis (see module docstring)."""
    tile = min(tile, length)
    if tile >= length:
        return [0]
    assert overlap < tile, "overlap must be smaller than the tile"
    n = int(np.ceil((length - overlap) / (tile - overlap)))
    return [int(round(i * (length - tile) / (n - 1))) for i in range(n)]


def calculate_overlap_rle(yranges, xranges, image_shape):
    """tile.py:8-52 -- RLE of the pixels covered by at least two tile rows or two tile columns."""
    y = np.array(rle_voting(np.unique(np.stack(yranges, axis=0), axis=0), vote_thr=2))
    x = np.array(rle_voting(np.unique(np.stack(xranges, axis=0), axis=0), vote_thr=2))
    if len(y) > 0:
        row_starts = y[:, 0] * image_shape[1]
        row_runs = y[:, 1] * image_shape[1] - row_starts
    else:
        row_starts, row_runs = [], []
    if len(x) > 0:
        col_ranges = np.concatenate([x + r * image_shape[1] for r in range(image_shape[0])], axis=0)
        col_starts = col_ranges[:, 0]
        col_runs = col_ranges[:, 1] - col_starts
    else:
        col_starts, col_runs = [], []
    if len(row_starts) > 0 and len(col_starts) > 0:
        return merge_rles(np.asarray(row_starts), np.asarray(row_runs), np.asarray(col_starts), np.asarray(col_runs))
    if len(row_starts) > 0:
        return merge_rles(np.asarray(row_starts), np.asarray(row_runs))
    if len(col_starts) > 0:
        return merge_rles(np.asarray(col_starts), np.asarray(col_runs))
    return [], []


class Tiler:
    """tile.py:54-194"""

    def __init__(self, image_shape, tile_size=2048, overlap_width=128):
        if isinstance(tile_size, int):
            tile_size = (tile_size, tile_size)
        assert isinstance(overlap_width, int)
        assert len(image_shape) == 2, "Tiler only works with 2D images"
        self.image_shape = image_shape
        self.tile_size = tile_size
        self.overlap_width = overlap_width
        th = min(tile_size[0], image_shape[0])
        tw = min(tile_size[1], image_shape[1])
        yranges, xranges = [], []
        for y in axis_offsets(image_shape[0], th, overlap_width):
            for x in axis_offsets(image_shape[1], tw, overlap_width):
                yranges.append((y, y + th))
                xranges.append((x, x + tw))
        self.overlap_rle = calculate_overlap_rle(yranges, xranges, image_shape)
        self.yranges = yranges
        self.xranges = xranges

    def __len__(self):
        return len(self.yranges)

    def overlap_mask(self):
        overlap = np.zeros(int(np.prod(self.image_shape)))
        for s, r in zip(self.overlap_rle[0], self.overlap_rle[1]):
            overlap[s:s + r] = 1
        return overlap.reshape(self.image_shape)

    def translate_rle_seg(self, rle_seg, tile_index):
        """tile.py:122-168 -- boxes and run starts from the tile frame to the image frame, in place."""
        ys, ye = self.yranges[tile_index]
        xs, xe = self.xranges[tile_index]
        w = xe - xs
        for labels in rle_seg.values():
            for attrs in labels.values():
                box = list(attrs['box'])
                box[0] += ys
                box[1] += xs
                box[2] += ys
                box[3] += xs
                attrs['box'] = tuple(box)
                starts = attrs['starts']
                attrs['starts'] = np.ravel_multi_index((starts // w + ys, starts % w + xs), dims=self.image_shape)
        return rle_seg

    def __call__(self, image, tile_index):
        if tile_index >= len(self):
            raise IndexError("Tile index out of range")
        assert image.shape == self.image_shape, \
            f"Image shape of {image.shape} does not match tiler expected shape {self.image_shape}"
        return image[slice(*self.yranges[tile_index]), slice(*self.xranges[tile_index])]
