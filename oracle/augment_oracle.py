"""CPU restatement of the reference's train-time augmentation maps -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

PARITY UNPINNED: the reference applies these maps through TensorFlow image ops (tf.image.crop_to_bounding_box,
tf.image.random_flip_left_right, tf.image.random_contrast), TensorFlow is not installed anywhere and the reference holds no
fixture for them; the restatement follows the call sites below and the published semantics of those ops.

    base()              annotator/data.py:195-206   centre crop, cast float32, / 255
    random_crop()       annotator/data.py:677-689   crop_to_bounding_box at (shape - output_size) // 2 + diff
    augment_random_flip annotator/data.py:620-625   left-right flip of the whole (features + label) image
    random_contrast()   annotator/data.py:586-609   tf.image.adjust_contrast: (x - mean_HW(x)) * factor + mean_HW(x) per channel,
                                                    on the target channels only, channel order restored
    to_feature_label()  annotator/data.py:766-788   label channel -> y, the others (in order) -> x
    random_warp()       annotator/data.py:725-763   tfa.image.sparse_image_warp(image, source, dest): tensorflow-addons is a
                                                    third-party dependency absent from /root/reference (requirements.txt:3,
                                                    unpinned); restated from its published algorithm: interpolate_spline
                                                    (order 2, phi(r) = 0.5 r log(max(r, 1e-10)) on squared distances, linear
                                                    term [q, 1] v, no regularisation) + dense_image_warp (bilinear sampling at
                                                    q - flow, floor clamped to [0, size - 2], alpha clipped to [0, 1])
"""

import numpy as np


def augment_image(img_u8, dy, dx, flip, contrast, output_size, label_index, target_channels=None):
    """One stored uint8 image [Hs, Ws, Cs] -> (x float32 [Ho, Wo, Cs-1], y float32 [Ho, Wo])."""
    hs, ws, cs = img_u8.shape
    ho, wo = output_size
    img = img_u8.astype(np.float32) / np.float32(255.0)                       # data.py:205-206
    top, left = (hs - ho) // 2 + dy, (ws - wo) // 2 + dx                      # data.py:683-687
    if top < 0 or left < 0 or top + ho > hs or left + wo > ws:
        raise ValueError('crop window leaves the image')                      # crop_to_bounding_box asserts
    img = img[top:top + ho, left:left + wo, :]
    if flip:
        img = img[:, ::-1, :]                                                 # flip_left_right
    if target_channels is None:
        target_channels = [c for c in range(cs) if c != label_index]
    out = img.copy()
    for c in target_channels:
        if c == label_index:
            continue
        mean = img[:, :, c].astype(np.float64).mean()                         # reduce_mean over H, W
        out[:, :, c] = ((img[:, :, c].astype(np.float64) - mean) * np.float64(np.float32(contrast)) + mean).astype(np.float32)
    feat = [c for c in range(cs) if c != label_index]
    return out[:, :, feat], out[:, :, label_index]


def augment_batch(raw_u8, params, output_size, label_index, target_channels=None):
    xs, ys = zip(*(augment_image(raw_u8[b], *params[b], output_size, label_index, target_channels) for b in range(len(raw_u8))))
    return np.stack(xs), np.stack(ys)


def warp_image(img, source, dest):
    """tfa.image.sparse_image_warp on one float image [H, W, C] (float64 arithmetic); source / dest [n, 2] (row, column)."""
    img = np.asarray(img, np.float64)
    h, w, _ = img.shape
    c = np.asarray(dest, np.float64)
    f = c - np.asarray(source, np.float64)
    k = len(c)
    phi = lambda r: 0.5 * r * np.log(np.maximum(r, 1e-10))                      # noqa: E731
    lhs = np.zeros((k + 3, k + 3))
    lhs[:k, :k] = phi(((c[:, None] - c[None]) ** 2).sum(-1))
    lhs[:k, k:k + 2] = c
    lhs[:k, k + 2] = 1.0
    lhs[k:, :k] = lhs[:k, k:].T
    rhs = np.zeros((k + 3, 2))
    rhs[:k] = f
    wv = np.linalg.solve(lhs, rhs)
    qy, qx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing='ij')
    q = np.stack([qy, qx], -1).reshape(-1, 2)
    flow = phi(((q[:, None] - c[None]) ** 2).sum(-1)) @ wv[:k] + np.concatenate([q, np.ones((len(q), 1))], 1) @ wv[k:]
    s = q - flow
    fy = np.clip(np.floor(s[:, 0]), 0, h - 2)
    fx = np.clip(np.floor(s[:, 1]), 0, w - 2)
    ay = np.clip(s[:, 0] - fy, 0, 1)[:, None]
    ax = np.clip(s[:, 1] - fx, 0, 1)[:, None]
    iy, ix = fy.astype(int), fx.astype(int)
    tl, tr, bl, br = img[iy, ix], img[iy, ix + 1], img[iy + 1, ix], img[iy + 1, ix + 1]
    top, bot = ax * (tr - tl) + tl, ax * (br - bl) + bl
    return (ay * (bot - top) + top).reshape(h, w, -1)
