"""Sliding-window tile grid of the scene (reference dataset.py:56-67)."""
import numpy as np


def get_patch_info_one_img(image_index, image_size, sample_margin, patch_size, patches_per_edge):
    """List of (image_index, (x0, y0), (x1, y1)); x outer / y inner; origins are the python-rounded
    points of linspace(margin, size - (patch + margin), n)."""
    lo = sample_margin
    hi = image_size - (patch_size + sample_margin)
    origins = [round(v) for v in np.linspace(start=lo, stop=hi, num=patches_per_edge)]
    return [(image_index, (x, y), (x + patch_size, y + patch_size)) for x in origins for y in origins]


def shard_tiles(n_tiles, world_size, rank):
    """Contiguous chunk of the tile list owned by `rank` (x-outer order makes each chunk a
    band of column strips).  Returns (begin, end)."""
    # balanced: chunk sizes differ by at most one and no rank is empty whenever n_tiles >= world_size
    return n_tiles * rank // world_size, n_tiles * (rank + 1) // world_size
