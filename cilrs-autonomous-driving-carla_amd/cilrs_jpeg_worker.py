"""JPEG decode worker for cilrs_mi355.data.BatchLoader's process pool.

Kept outside the package on purpose: a spawned worker imports only numpy and Pillow (not torch or
the HIP library), so a pool of 16 starts in about a second and never touches the GPU.
"""
import numpy as np
from PIL import Image


def decode_chunk(args):
    """(paths, height, width) -> uint8 [n, height, width, 3] RGB (cv2.imread + BGR2RGB in the
    reference, notebook/notebook.ipynb:408-409)."""
    paths, h, w = args
    out = np.empty((len(paths), h, w, 3), dtype=np.uint8)
    for k, p in enumerate(paths):
        with Image.open(p) as im:
            if im.mode != "RGB":
                im = im.convert("RGB")
            a = np.asarray(im)
        if a.shape != (h, w, 3):
            raise RuntimeError(f"{p}: expected {w}x{h} RGB, got {a.shape}")
        out[k] = a
    return out
