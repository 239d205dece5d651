"""The bf16-emulating oracle comparison of tests/test_engine_gpu.py as a stand-alone program, so that it can run in a child
process with DNNCA_IGB_NW=8 (the variant of k_igb_conv3 -- eight waves, 32 x 16-pixel tiles -- that the unet_big benchmark
runs; the library reads the variable once per process).  Prints one JSON line.

dtype bf16: the implicit-GEMM kernels round their operands (activations, gradients, weights) to bf16 while staging and
accumulate in fp32.  The oracle is made to do exactly that (operands of every 3x3 conv and every 64-multiple transposed conv
rounded to bf16, float64 accumulation), so the comparison isolates the kernels' indexing from bf16 noise.
With BatchNorm ("bf16 activations", BASELINE.md configs[2]) the outputs of the 64-channel-multiple convs / transposed convs
that feed a BatchNorm are stored as bf16 as well (ig_plan_half): the oracle rounds those outputs too (round_z).
With BatchNorm the gradients that arrive at a BatchNorm from the 64-channel kernels are stored as bf16 (round_dy).
"""

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

import helpers as Hp                      # noqa: E402
from oracle import unet_oracle as O       # noqa: E402


def to_bf16(a):
    """round-to-nearest-even to bfloat16 precision (what v_cvt_pk_bf16_f32 does), returned as float64"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


class bf16_oracle:
    """context manager: every 3x3 conv / 64-multiple transposed conv of the oracle contracts bf16-rounded operands"""

    def __init__(self, round_z=False):
        self.round_z = round_z

    def __enter__(self):
        round_z = self.round_z
        self.saved = (O.conv2d_fwd, O.conv2d_bwd, O.tconv_fwd, O.tconv_bwd)
        self.saved_bn = O.bn_bwd
        fwd0, bwd0, tfwd0, tbwd0 = self.saved
        bn_bwd0 = O.bn_bwd
        calls = [0]

        def bn_bwd(cache, dy):
            # the gradient arriving at a BatchNorm whose users are all 64-channel bf16 kernels is stored as bf16: in these
            # networks every BatchNorm but the last one (the first whose backward runs: it feeds the head)
            calls[0] += 1
            if round_z and calls[0] > 1 and dy.shape[-1] % 64 == 0:
                dy = to_bf16(dy)
            return bn_bwd0(cache, dy)

        O.bn_bwd = bn_bwd

        def tc_bf16(w):
            return w.shape[2] % 64 == 0 and w.shape[3] % 64 == 0

        def tfwd(x, w, b):
            if not tc_bf16(w):
                return tfwd0(x, w, b)
            out = tfwd0(to_bf16(x), to_bf16(w), b)
            if round_z:                      # (y, cache): the BatchNorm behind it reads the stored (rounded) values
                out = (to_bf16(out[0]),) + tuple(out[1:])
            return out

        def z_half(w):         # conv_ok() of ig_plan_half: every source and the output a multiple of 64 channels
            cin, cout = w.shape[2], w.shape[3]
            return cout % 64 == 0 and (cin == 2 * cout or cin % 64 == 0)      # cin == 2 cout: the decoder's two-source conv

        def tbwd(cache, dy):
            return tbwd0(cache, to_bf16(dy)) if tc_bf16(cache[1]) else tbwd0(cache, dy)

        def dense(w):          # use_bf16() of csrc/kernels_igemm.hip: 3x3, every source and the output a multiple of 32 channels
            return w.shape[0] == 3 and w.shape[2] % 32 == 0 and w.shape[3] % 32 == 0

        def fwd(x, w, b, padding, alpha=None):
            if not dense(w):
                return fwd0(x, w, b, padding, alpha)
            if not (round_z and z_half(w)):
                return fwd0(to_bf16(x), to_bf16(w), b, padding, alpha)
            y, cache = fwd0(to_bf16(x), to_bf16(w), b, padding, alpha)
            y = to_bf16(y)
            xp, wc, yv, al, pad, xshape = cache
            return y, (xp, wc, y, al, pad, xshape)       # act' in the backward pass sees the stored values

        def bwd(cache, dy):
            xp, w, yv, alpha, padding, xshape = cache
            if not dense(w):
                return bwd0(cache, dy)
            dz = dy if alpha is None else O._act_bwd(yv, dy, alpha)
            return bwd0((xp, w, yv, None, padding, xshape), to_bf16(dz))      # x and w in the cache are already rounded

        O.conv2d_fwd, O.conv2d_bwd, O.tconv_fwd, O.tconv_bwd = fwd, bwd, tfwd, tbwd
        return self

    def __exit__(self, *exc):
        O.conv2d_fwd, O.conv2d_bwd, O.tconv_fwd, O.tconv_bwd = self.saved
        O.bn_bwd = self.saved_bn


def run(device, f0, S, B=2, bn=False, cin=32, n_down=2):
    """Network: every 3x3 conv has f0..2*f0 channels (all on the bf16 path: f0 = 32 runs the 16/32-channel-tile kernels,
    f0 = 64 the 64-channel-tile ones); transposed convs whose channel counts are multiples of 64 contract in bf16 as well."""
    opts = dict(rate=2, kernel_size=3, conv_stride=1, padding='same', n_filters_first=f0, n_downsample=n_down, bn=bn)
    spec = O.ModelSpec('unet', cin, **opts)
    params = Hp.perturbed_params(spec, np.float64)
    rng = np.random.default_rng(3)
    x = rng.random((B, S, S, cin)).astype(np.float32)
    _, y = O.synthetic_batch(B, S, S, 1)
    cfg = dict(weight_mul=3.0)
    with bf16_oracle(round_z=bn):
        loss, grads, logits, _ = O.loss_and_grads(spec, params, x.astype(np.float64), y, cfg, training=True)
    m = device.DeviceModel('unet', cin, S, S, B, dtype='bf16', **opts)
    m.set_params(O.flatten(spec, params))
    if m.n_state:
        m.set_state(O.flatten(spec, params, trainable=False))
    _, lg = m.forward(x, training=False, return_logits=True)
    dl = np.abs(lg - logits)
    out = m.train_step(x, y, 0.0, m.loss_cfg(**cfg))
    g, gref = m.get_grads().astype(np.float64), O.flatten(spec, grads)
    errs = Hp.per_tensor_err(spec, g, gref)
    names = sorted(set(r[0] for r in m.plan()) | set(r[0] for r in m.plan(variants=True)))
    m.close()
    deg = Hp.degenerate_tensors(spec)
    errs = {n: e for n, e in errs.items() if n not in deg}
    return dict(f0=f0, S=S, dl_max=float(dl.max()), dl_median=float(np.median(dl)), loss=float(out.loss), loss_ref=float(loss),
                err_l2=float(np.linalg.norm(g - gref) / np.linalg.norm(gref)), per_tensor=errs, plan=names)


if __name__ == '__main__':
    from dnncancerannotator_amd import device as dev
    dev.init_device(0)
    a = [int(v) for v in sys.argv[1:]]
    print(json.dumps(run(dev, a[0], a[1], *(a[2:3] or [2]), bn=bool(a[3]) if len(a) > 3 else False, cin=a[4] if len(a) > 4 else 32,
                         n_down=a[5] if len(a) > 5 else 2)))
