"""Synthetic MRI-shaped batches: x = uint8/255 like annotator/data.py:205-206, y = union of 0-3 discs."""

import numpy as np


def synthetic_batch(B, H, W, C, seed_x=0, seed_y=1):
    rx = np.random.default_rng(seed_x)
    ry = np.random.default_rng(seed_y)
    x = (rx.integers(0, 256, size=(B, H, W, C), dtype=np.uint8) / np.float32(255.0)).astype(np.float32)
    y = np.zeros((B, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    s = min(H, W) / 512.0
    for b in range(B):
        n = int(ry.integers(0, 4))
        if b == 0:
            n = max(n, 1)
        for _ in range(n):
            r = ry.uniform(8, 40) * s
            cy, cx = ry.uniform(64, 448, 2) * s
            y[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1.0
    return x, y
