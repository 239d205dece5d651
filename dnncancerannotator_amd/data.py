"""Datasets for the engine.  The element contract is the reference's (annotator/data.py:193,206,766-788):
batches (x float32 [B,H,W,C] in [0,1], y float32 [B,H,W]).  Any iterable of such tuples works; these helpers add the
`element_spec` the engine reads the input shape from (engine.py:93)."""

from collections import namedtuple

import numpy as np

from .synthetic import synthetic_batch

Spec = namedtuple('Spec', ['shape', 'dtype'])


class ArrayDataset:
    """Batches over in-memory arrays; `repeat=True` makes it endless like the reference's train_ds (data.py:62-111)."""

    def __init__(self, x, y, batch_size, repeat=False, drop_remainder=False):
        self.x = np.ascontiguousarray(x, np.float32)
        self.y = np.ascontiguousarray(y, np.float32)
        assert self.x.ndim == 4 and self.y.shape == self.x.shape[:3]
        self.batch_size, self.repeat, self.drop_remainder = int(batch_size), repeat, drop_remainder
        self.element_spec = (Spec((self.batch_size,) + self.x.shape[1:], np.float32),
                             Spec((self.batch_size,) + self.y.shape[1:], np.float32))

    def __iter__(self):
        n = len(self.x)
        while True:
            for i in range(0, n, self.batch_size):
                if i + self.batch_size > n and self.drop_remainder:
                    break
                yield self.x[i:i + self.batch_size], self.y[i:i + self.batch_size]
            if not self.repeat:
                return


class SyntheticDataset(ArrayDataset):
    """Endless synthetic MRI-shaped batches: `n_batches` distinct batches, cycled."""

    def __init__(self, batch_size, height=512, width=512, channels=1, n_batches=4, seed=0, repeat=True):
        xs, ys = [], []
        for i in range(n_batches):
            x, y = synthetic_batch(batch_size, height, width, channels, seed_x=seed + 2 * i, seed_y=seed + 2 * i + 1)
            xs.append(x)
            ys.append(y)
        super().__init__(np.concatenate(xs), np.concatenate(ys), batch_size, repeat=repeat)
