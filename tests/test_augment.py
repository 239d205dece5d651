"""Train-time augmentation: the host-side draws and option parsing on CPU, the device kernels against the
oracle restatement on the GPU, and the engine training on augmented uint8 batches end to end."""

import os

import numpy as np
import pytest

from oracle import augment_oracle as A


def test_parse_and_draws():
    from dnncancerannotator_amd import augment
    # configs/additionals/data_options.yaml:9-13 (every option empty); train_ds fills in the defaults (data.py:87-93)
    plan = augment.parse_augment_options({'random_crop': None, 'random_flip': None, 'random_contrast': None, 'random_warp': None}, (256, 256))
    assert plan.crop == dict(stddev=4, max_=6, min_=-6) and plan.flip and plan.contrast['lower'] == 0.8 and plan.output_size == (256, 256)
    assert plan.warp == dict(n_points=100, max_diff=5, stddev=2.0)
    assert augment.parse_augment_options(None, (128, 128)).crop is not None          # train_ds: at least the random crop
    with pytest.raises(KeyError):
        augment.parse_augment_options({'random_hue2': {}}, (8, 8))
    rng = np.random.default_rng(0)
    draws = augment.draw_params(rng, 4000, plan)
    dy = np.array([d[0] for d in draws])
    assert dy.min() >= -6 and dy.max() <= 6 and abs(dy.mean()) < 0.3 and 2.5 < dy.std() < 4.5      # int(N(0,4)) clipped to [-6, 6]
    assert 0.45 < np.mean([d[2] for d in draws]) < 0.55
    f = np.array([d[3] for d in draws])
    assert f.min() >= 0.8 and f.max() < 1.2
    none = augment.draw_params(rng, 3, augment.parse_augment_options({}, (8, 8)))
    assert none == [(0, 0, 0, 1.0)] * 3


def test_oracle_properties():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (20, 24, 4), np.uint8)
    x, y = A.augment_image(img, 0, 0, 0, 1.0, (20, 24), 3)
    assert np.array_equal(x, img[..., :3].astype(np.float32) / np.float32(255)) and np.array_equal(y, img[..., 3].astype(np.float32) / np.float32(255))
    x2, y2 = A.augment_image(img, 1, -2, 1, 1.0, (12, 16), 3)
    ref = (img[4 + 1:4 + 1 + 12, 4 - 2:4 - 2 + 16][:, ::-1].astype(np.float32) / np.float32(255))
    assert np.array_equal(x2, ref[..., :3]) and np.array_equal(y2, ref[..., 3])
    x3, y3 = A.augment_image(img, 0, 0, 0, 1.2, (12, 16), 3)
    base, _ = A.augment_image(img, 0, 0, 0, 1.0, (12, 16), 3)
    assert np.allclose(x3.mean((0, 1)), base.mean((0, 1)), atol=1e-6)                 # contrast keeps the channel means
    assert np.allclose(x3 - base.mean((0, 1)), 1.2 * (base - base.mean((0, 1))), atol=1e-6)
    assert np.array_equal(y3, A.augment_image(img, 0, 0, 0, 1.0, (12, 16), 3)[1])     # the label is never adjusted
    with pytest.raises(ValueError):
        A.augment_image(img, 5, 0, 0, 1.0, (12, 16), 3)


@pytest.mark.gpu
@pytest.mark.parametrize('cs,label_index', [(6, 5), (2, 1), (4, 1)])
def test_device_augment_matches_oracle(gpu, cs, label_index):
    from dnncancerannotator_amd import augment
    rng = np.random.default_rng(2)
    B, hs, ws, ho, wo = 5, 96, 80, 64, 48
    raw = rng.integers(0, 256, (B, hs, ws, cs), np.uint8)
    plan = augment.parse_augment_options({'random_crop': {'stddev': 4}, 'random_flip': {}, 'random_contrast': {}}, (ho, wo))
    params = augment.draw_params(rng, B, plan)
    params[0] = (6, -6, 1, 1.2)                                   # extremes
    m = gpu.DeviceModel('unet', cs - 1, ho, wo, B, n_filters_first=3, n_downsample=1, rate=2, kernel_size=3, conv_stride=1, padding='same')
    xb, yb = m.augment_u8(raw, params, (ho, wo), label_index)
    xr, yr = A.augment_batch(raw, params, (ho, wo), label_index)
    assert np.array_equal(yb.to_host(), yr)                       # crop / flip / 255 are exact
    assert np.abs(xb.to_host() - xr).max() <= 2e-6                # contrast: float32 vs float64 mean
    no_c = [(p[0], p[1], p[2], 1.0) for p in params]
    xb, _ = m.augment_u8(raw, no_c, (ho, wo), label_index)
    assert np.array_equal(xb.to_host(), A.augment_batch(raw, no_c, (ho, wo), label_index)[0])
    with pytest.raises(RuntimeError):
        m.augment_u8(raw, [(17, 0, 0, 1.0)] * B, (ho, wo), label_index)          # window leaves the image
    m.close()


def test_warp_solution_interpolates_the_control_flows():
    """host side of random_warp: the spline solved by augment.solve_warp reproduces the control-point flows at the control
    points (interpolation property) and agrees with the oracle's own solve through the warped image of a smooth ramp."""
    from dnncancerannotator_amd import augment
    rng = np.random.default_rng(3)
    src, dst = augment.draw_warp(rng, 2, 32, n_points=20)
    assert src.shape == (2, 20, 2) and np.abs(dst - src).max() <= 5.0 and src.min() >= 0 and src.max() < 32
    ctrl, wv = augment.solve_warp(src, dst)
    for b in range(2):
        c = ctrl[b].astype(np.float64)
        d2 = ((c[:, None] - c[None]) ** 2).sum(-1)
        flow = (0.5 * d2 * np.log(np.maximum(d2, 1e-10))) @ wv[b, :20] + np.concatenate([c, np.ones((20, 1))], 1) @ wv[b, 20:]
        assert np.abs(flow - (dst[b] - src[b])).max() < 1e-3
    ident = A.warp_image(rng.random((16, 16, 2)), src[0, :5] / 2, src[0, :5] / 2)           # zero flow: identity
    assert ident.shape == (16, 16, 2)


@pytest.mark.gpu
def test_device_warp_matches_oracle(gpu):
    """dnnca_warp_f32 against the oracle's restatement of tfa.image.sparse_image_warp.  Channel 0 / 1 of the warped image are
    the row / column ramps, so their output IS the sampling position q - flow(q): it must agree to a hundredth of a pixel; the
    third channel and the label are smooth images (a noise image would amplify that hundredth by its gradients)."""
    from dnncancerannotator_amd import augment
    rng = np.random.default_rng(4)
    B, S = 3, 48
    yy, xx = np.mgrid[:S, :S].astype(np.float32)
    smooth = 0.5 + 0.25 * np.sin(yy / 7.0) * np.cos(xx / 5.0)
    x0 = np.stack([np.stack([yy / S, xx / S, smooth], -1)] * B).astype(np.float32)
    y0 = np.stack([(smooth > 0.55).astype(np.float32) * 0 + smooth[::-1]] * B).astype(np.float32)
    m = gpu.DeviceModel('unet', 3, S, S, B, n_filters_first=3, n_downsample=1, rate=2, kernel_size=3, conv_stride=1, padding='same')
    src, dst = augment.draw_warp(rng, B, S, n_points=100)
    xw, yw = m.warp(gpu.DeviceBuffer(x0), gpu.DeviceBuffer(y0), *augment.solve_warp(src, dst))
    xw, yw = xw.to_host(), yw.to_host()
    for b in range(B):
        ref = A.warp_image(np.concatenate([x0[b], y0[b][..., None]], -1), src[b], dst[b])
        assert np.abs(xw[b][..., :2] - ref[..., :2]).max() * S <= 1e-2            # sampling positions, in pixels
        assert np.abs(xw[b][..., 2] - ref[..., 2]).max() <= 1e-3 and np.abs(yw[b] - ref[..., 3]).max() <= 1e-3
        assert np.abs(xw[b][..., :2] - x0[b][..., :2]).max() * S > 1.0             # and the image did move (by pixels)
    m.close()


@pytest.mark.gpu
def test_engine_trains_on_augmented_tfrecords(gpu, tmp_path):
    """`annotator train`'s dataset for .tfrecords with the reference's augment_options: uint8 batches + draws from the dataset,
    augmentation + train step on the device; the loss goes down on a learnable synthetic exam."""
    from dnncancerannotator_amd import engine, tfrecord
    from dnncancerannotator_amd.runs.train import make_dataset
    rng = np.random.default_rng(5)
    n, s = 12, 80
    label = np.zeros((n, s, s), np.uint8)
    yy, xx = np.mgrid[:s, :s]
    for k in range(n):
        cy, cx = rng.integers(24, s - 24, 2)
        label[k][(yy - cy) ** 2 + (xx - cx) ** 2 < 100] = 255
    tra = (label * 0.6 + rng.integers(0, 90, label.shape)).astype(np.uint8)          # the lesion is visible in the feature channel
    slices = np.stack([tra, rng.integers(0, 256, label.shape).astype(np.uint8), label], -1)
    rec = str(tmp_path / 'exam.tfrecords')
    tfrecord.write_records(rec, [tfrecord.make_example(slices, 1, 1, 'p', 'cancer', ['TRA', 'ADC', 'label'])])
    opts = dict(batch_size=4, buffer_size=8, output_size=[64, 64], slice_types=['TRA', 'ADC', 'label'],
                augment_options={'random_crop': None, 'random_flip': None, 'random_contrast': None, 'random_warp': None})
    ds = make_dataset([rec], opts, training=True)
    assert ds.element_spec[0].shape == (4, 64, 64, 2)
    cfg = {'model': 'UNetAnnotator',
           'model_options': dict(n_filters_first=3, n_downsample=2, rate=2, kernel_size=3, conv_stride=1, bn=False, padding='same'),
           'deploy_options': {'optimizer': 'adam', 'loss': {'class_name': 'WeightedCrossentropy', 'config': {'weight_mul': 3.0}},
                              'enable_multigpu': False}}
    m = engine.TFKerasModel(cfg)
    res = m.train(ds, save_path=str(tmp_path / 'run'), max_steps=60, save_freq=1000)
    loss = res.history['loss']
    assert len(loss) == 60 and np.mean(loss[-10:]) < 0.8 * np.mean(loss[:5])
