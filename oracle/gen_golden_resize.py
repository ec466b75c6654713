#!/usr/bin/env python3
"""tests/golden/pil_resize_golden.npz: outputs of Pillow itself -- ``Image.fromarray(img).resize((nw, nh), BILINEAR)``,
the call detectron2's ResizeTransform makes for uint8 images -- on small seeded images (up-, down- and mixed scaling),
so that oracle/rcnn_ref.pil_resize_bilinear_u8 stays pinned where Pillow is not installed.
Run here: python oracle/gen_golden_resize.py"""
import os

import numpy as np
import PIL
from PIL import Image

CASES = [(48, 64, 80, 107), (60, 45, 86, 64), (90, 120, 40, 53), (33, 47, 70, 31), (37, 53, 37, 90)]


def main():
    out = {"pillow_version": np.array(PIL.__version__)}
    rng = np.random.RandomState(2024)
    for i, (h, w, nh, nw) in enumerate(CASES):
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        out[f"in{i}"] = img
        out[f"out{i}"] = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pil_resize_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
